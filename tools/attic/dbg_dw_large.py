import os, sys
import torch
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from xmm_superres_denoise.engine import Engine
from xmm_superres_denoise.engine._lib import check
from util_hip import ptr_array
for (B, H, W, n_in) in [(2, 64, 64, 3), (16, 512, 512, 3), (16, 512, 512, 1), (4, 512, 512, 3)]:
    gen = torch.Generator(device="cuda").manual_seed(5)
    e = Engine("dn", 1, 1, 32, 1); e.set_math("f16x3")
    xs = [torch.randn((B, H, W, 32), device="cuda", generator=gen) for _ in range(n_in)]
    g = torch.randn((B, H, W, 32), device="cuda", generator=gen)
    w = torch.randn((32, 32 * n_in, 3, 3), device="cuda", generator=gen) / (288 * n_in) ** 0.5
    dxs = [torch.empty((B, H, W, 32), device="cuda") for _ in range(n_in)]
    dw, db = torch.full_like(w, float("nan")), torch.full((32,), float("nan"), device="cuda")
    check(e.L.xsd_test_conv3x3_bwd(e.h, ptr_array(xs), n_in, w.data_ptr(), g.data_ptr(), ptr_array(dxs), dw.data_ptr(), db.data_ptr(), B, H, W, None))
    torch.cuda.synchronize()
    print((B, H, W, n_in), "nonfinite dw", int((~torch.isfinite(dw)).sum()), "of", dw.numel(), "db", int((~torch.isfinite(db)).sum()), "max|dw|", float(dw[torch.isfinite(dw)].abs().max()) if torch.isfinite(dw).any() else None)
