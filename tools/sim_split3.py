"""CPU simulation (numerics only, no GPU): whole-net error of the 3-term bf16 split with 6 / 8 / 9 products versus a float64
evaluation, next to torch's own fp32 path.  Products and sums of the split terms are evaluated in float64 here, so the
number isolates the error of DROPPING the low-order cross terms (the accumulate error is the MFMA's fp32 accumulator, the
same as any fp32 path)."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import gen_common as gc
from oracle import oracle

def bf(t):  # round-to-nearest-even bf16 of a float32 tensor, returned as float32
    return t.to(torch.bfloat16).to(torch.float32)

def split3(t32):
    h = bf(t32); r = t32 - h; m = bf(r); l = bf(r - m)
    return h.double(), m.double(), l.double()

def make_conv(nprod, state):
    def conv(name, t):
        w, b = state[name + ".weight"], state[name + ".bias"]
        if nprod == 0:
            return F.conv2d(t, w, b, padding=1)
        t32 = t.float()      # planes are fp32 in HBM
        if w.shape[1] == 1 or w.shape[0] == 1:   # edge layers are VALU fp32 kernels: exact products
            return F.conv2d(t32.double(), w.double(), b.double(), padding=1)
        xh, xm, xl = split3(t32); wh, wm, wl = split3(w.float())
        terms = [(wh, xh), (wh, xm), (wm, xh), (wh, xl), (wl, xh), (wm, xm)]
        if nprod >= 8: terms += [(wm, xl), (wl, xm)]
        if nprod >= 9: terms += [(wl, xl)]
        out = b.double().view(1, -1, 1, 1)
        for ww, xx in terms: out = out + F.conv2d(xx, ww, None, padding=1)
        return out.float().double()   # fp32 plane store
    return conv

def forward(kind, blocks, state, x, conv, dt):
    fea = conv("conv_first", x); cur = fea
    for i in range(blocks):
        rin = cur
        for r in (1, 2, 3):
            pre = f"rrdb.{i}.RDB{r}."; xs = [cur]
            for c in (1, 2, 3, 4): xs.append(F.leaky_relu(conv(pre + f"conv{c}", torch.cat(xs, 1)), 0.2))
            cur = conv(pre + "conv5", torch.cat(xs, 1)) * 0.2 + cur
        cur = cur * 0.2 + rin
    fea = fea + conv("trunk_conv", cur)
    out = conv("conv_last", fea) + x
    return out  # pre-clamp (so that the error is visible everywhere)

torch.set_num_threads(8)
blocks, n = 4, 96
for seed in (11, 12):
    st = gc.make_state("dn", 32, blocks, seed)
    x = torch.from_numpy(gc.make_input((1, 1, n, n), seed + 100))
    s64 = {k: torch.from_numpy(v).double() for k, v in st.items()}
    s32 = {k: torch.from_numpy(v) for k, v in st.items()}
    y64 = forward("dn", blocks, s64, x.double(), make_conv(0, s64), torch.float64)
    y32 = forward("dn", blocks, s32, x, make_conv(0, s32), torch.float32).double()
    res = {"torch fp32": y32}
    for npd in (6, 8):
        res[f"split3 x{npd}"] = forward("dn", blocks, s32, x.double(), make_conv(npd, s32), torch.float64)
    sc = y64.abs().max().item()
    for k, v in res.items():
        e = (v - y64).abs()
        print(f"seed {seed} {k:12s} max|err|/max|y| {e.max().item()/sc:.3e}  rms {e.pow(2).mean().sqrt().item()/sc:.3e}")
