"""Where does dL/dx of the SR generator deviate from a float64 evaluation?  (debug helper; usage: python tools/dbg_sr_dx.py [size] [blocks])"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/xmm-superres-denoise_amd", ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import gen_common as gc
from oracle import oracle
from util_hip import build_module
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = 7701
state = gc.make_state("sr", 32, blocks, seed, gain=1.0, last_bias=0.3)
x = gc.make_input((1, 1, size, size), seed + 1)
dy = (gc.make_input((1, 1, 2 * size, 2 * size), seed + 2) - 0.5).astype(np.float32) / (4 * size * size)
st = {k: torch.from_numpy(v).double().cuda().requires_grad_(True) for k, v in state.items()}
xt = torch.from_numpy(x).double().cuda().requires_grad_(True)
y = oracle.torch_forward("sr", 32, blocks, st, xt)
y.backward(torch.from_numpy(dy).double().cuda())
dx64 = xt.grad.cpu().numpy()[0, 0]
g64 = {k: v.grad.cpu().numpy() for k, v in st.items()}
for math in ("fp32", "bf16x6", "f16x3"):
    m = build_module("sr", blocks, 1, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    yy = eng.forward(torch.from_numpy(x).cuda(), save_for_backward=True)
    grads = torch.empty_like(m.flat_parameters())
    dx = eng.backward(torch.from_numpy(dy).cuda(), grads, need_dx=True).cpu().numpy()[0, 0]
    err = np.abs(dx - dx64) / np.abs(dx64).max()
    rows = np.nonzero(err.max(axis=1) > 4e-4)[0]; cols = np.nonzero(err.max(axis=0) > 4e-4)[0]
    print(math, "max rel err %.3e at %s; rows over 4e-4: %s cols: %s; |y - y64| max %.2e" % (err.max(), np.unravel_index(err.argmax(), err.shape), rows[:20], cols[:20], np.abs(yy.cpu().numpy() - y.detach().cpu().numpy()).max()))
    off = 0
    worst = []
    for n, shp in gc.rrdb_param_shapes("sr", 32, blocks).items():
        k = int(np.prod(shp)); g = grads[off:off + k].cpu().numpy().reshape(shp); off += k
        e = np.abs(g - g64[n]).max() / (np.abs(g64[n]).max() + 1e-30)
        worst.append((e, n))
    print("   worst parameter-gradient tensors:", sorted(worst)[-4:])
