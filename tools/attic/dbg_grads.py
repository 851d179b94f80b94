"""Diagnostic: per-parameter gradient error of each math mode against the C oracle and against each other on one of the
test_fresh_inputs_vs_oracle cases.  usage: python tools/dbg_grads.py sr 2 19 22 2"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gen_common as gc
from oracle import oracle
from util_hip import build_module
kind, B, H, W, nup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
shape = (B, 1, H, W)
state = gc.make_state(kind, 32, 2, 900, num_upsample=nup, last_bias=0.3 if kind == "sr" else None)
x = gc.make_input(shape, 901)
s = 2 ** nup if kind == "sr" else 1
t = gc.make_input((shape[0], 1, shape[2] * s, shape[3] * s), 902)
yo, lo, dxo, go = oracle.l1_train(kind, 32, 2, oracle.flatten_state(state), x, t, num_upsample=nup)
# float64 torch reference
st64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in state.items()}
x64 = torch.from_numpy(x).double().requires_grad_(True)
y64 = oracle.torch_forward(kind, 32, 2, st64, x64, num_upsample=nup)
torch.nn.functional.l1_loss(y64, torch.from_numpy(t).double()).backward()
g64 = torch.cat([v.grad.reshape(-1) for v in st64.values()]).numpy()
res = {"oracle": (yo, go, dxo)}
for math in ("fp32", "bf16x6", "f16x3"):
    m = build_module(kind, 2, nup, state).set_math(math)
    eng = m._get_engine(torch.device("cuda", 0))
    eng.pack(m.flat_parameters())
    y = eng.forward(torch.from_numpy(x).cuda(), save_for_backward=True)
    loss, dy = eng.l1_loss(y, torch.from_numpy(t).cuda())
    grads = torch.empty_like(m.flat_parameters())
    dx = eng.backward(dy, grads, need_dx=True)
    res[math] = (y.cpu().numpy(), grads.cpu().numpy(), dx.cpu().numpy())
shapes = gc.rrdb_param_shapes(kind, 32, 2, num_upsample=nup)
print("sign(y-t) differences vs float64:", {k: int((np.sign(v[0] - t) != np.sign(y64.detach().numpy() - t)).sum()) for k, v in res.items()})
off = 0
print(f"{'param':34s}" + "".join(f"{k:>12s}" for k in res) + "   (max |g - g64| / max|g64|)")
for n, shp in shapes.items():
    k = int(np.prod(shp))
    ref = g64[off:off + k]
    sc = np.abs(ref).max() + 1e-300
    print(f"{n:34s}" + "".join(f"{np.abs(v[1][off:off + k] - ref).max() / sc:12.2e}" for v in res.values()))
    off += k
