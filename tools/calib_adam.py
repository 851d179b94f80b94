"""Calibration helper for tests/test_hip_network.py::test_train_step_adam_matches_oracle: accumulated Adam update error of
each fp32-class math mode against the oracle, as a function of how far above the largest gradient's 1e-k a parameter's
gradient has to stay to count as "solid".  usage (GPU box): python tools/calib_adam.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/xmm-superres-denoise_amd", ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import gen_common as gc
from oracle import oracle
from util_hip import build_module
for math in ("fp32", "bf16x6", "f16x3"):
    kind, blocks = "dn", 1
    state = gc.make_state(kind, 32, blocks, 77)
    x = gc.make_input((2, 1, 24, 40), 78); t = gc.make_input((2, 1, 24, 40), 79)
    p = oracle.flatten_state(state).copy(); mo, vo = np.zeros_like(p), np.zeros_like(p)
    m = build_module(kind, blocks, 1, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    flat = m.flat_parameters(); md, vd = torch.zeros_like(flat), torch.zeros_like(flat)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    grads = torch.empty_like(flat); g_min = None
    for step in range(1, 4):
        _, lo, _, go = oracle.l1_train(kind, 32, blocks, p, x, t)
        g_min = np.abs(go) if g_min is None else np.minimum(g_min, np.abs(go))
        oracle.adam(p, go, mo, vo, step)
        eng.pack(flat); y = eng.forward(xd, save_for_backward=True); loss, dy = eng.l1_loss(y, td)
        eng.backward(dy, grads); eng.adam_step(flat, grads, md, vd, step)
    start = oracle.flatten_state(state)
    d_eng = flat.cpu().numpy().astype(np.float64) - start; d_ora = p.astype(np.float64) - start
    gerr = np.abs(grads.cpu().numpy().astype(np.float64) - go).max() / np.abs(go).max()
    print(math, "last-step max |g - g_oracle| / max|g| = %.2e" % gerr)
    for k in (6, 5, 4, 3):
        solid = g_min > 10.0 ** -k * np.abs(go).max()
        print("   solid > 1e-%d max: %.1f%% of parameters, max update diff %.2e" % (k, 100 * solid.mean(), np.abs(d_eng[solid] - d_ora[solid]).max()))
