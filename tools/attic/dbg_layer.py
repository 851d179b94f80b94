"""Diagnostic: single conv layer forward / input-gradient / weight-gradient rms error vs float64 per math mode."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from util_hip import nchw_to_planes, planes_to_nchw, ptr_array
from xmm_superres_denoise.engine import Engine
from xmm_superres_denoise.engine._lib import check
def rms(a, r):
    a, r = np.asarray(a, np.float64), np.asarray(r, np.float64)
    return float(np.sqrt(np.mean((a - r) ** 2)) / np.sqrt(np.mean(r ** 2)))
for (n_in, B, H, W, gscale) in [(1, 2, 64, 64, 1.0), (5, 1, 32, 64, 1.0), (3, 2, 40, 72, 1e-6)]:
    rng = np.random.default_rng(7 * n_in)
    x = rng.normal(size=(B, 32 * n_in, H, W)).astype(np.float32)
    w = (rng.normal(size=(32, 32 * n_in, 3, 3)) / np.sqrt(288 * n_in)).astype(np.float32)
    g = (rng.normal(size=(B, 32, H, W)) * gscale).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True); wt = torch.from_numpy(w).double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, wt, None, padding=1); y.backward(torch.from_numpy(g).double())
    dx64, dw64, db64 = xt.grad.numpy(), wt.grad.numpy(), g.astype(np.float64).sum(axis=(0, 2, 3))
    xt32 = torch.from_numpy(x).requires_grad_(True); wt32 = torch.from_numpy(w).requires_grad_(True)
    torch.nn.functional.conv2d(xt32, wt32, None, padding=1).backward(torch.from_numpy(g))
    print(f"n_in={n_in} B={B} {H}x{W} gscale={gscale}: torch fp32 dx {rms(xt32.grad.numpy(), dx64):.2e} dw {rms(wt32.grad.numpy(), dw64):.2e}")
    xin = nchw_to_planes(x); gp = nchw_to_planes(g)[0]
    wd = torch.from_numpy(w).cuda()
    for math in ("fp32", "bf16x6", "f16x3"):
        e = Engine("dn", 1, 1, 32, 1); e.set_math(math)
        dxs = [torch.full((B, H, W, 32), float("nan"), device="cuda") for _ in range(n_in)]
        dw = torch.full_like(wd, float("nan")); db = torch.full((32,), float("nan"), device="cuda")
        check(e.L.xsd_test_conv3x3_bwd(e.h, ptr_array(xin), n_in, wd.data_ptr(), gp.data_ptr(), ptr_array(dxs), dw.data_ptr(), db.data_ptr(), B, H, W, None))
        print(f"   {math:8s} dx {rms(planes_to_nchw(dxs), dx64):.2e}  dw {rms(dw.cpu().numpy(), dw64):.2e}  db {rms(db.cpu().numpy(), db64):.2e}")
