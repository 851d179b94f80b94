"""64-filter DN net at full size (512 x 512, batch 2, 4 blocks): the three math modes of the plane kernels against each other
(forward output and the flat parameter gradient of an L1 train step).  Usage (through gpurun): python tools/wide_fullsize_check.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xmm-superres-denoise_amd"))
from xmm_superres_denoise.models import GeneratorRRDB_DN
from xmm_superres_denoise.parallel import DataParallelTrainer
torch.manual_seed(0)
m = GeneratorRRDB_DN(1, 1, 64, 4).cuda()
x = torch.rand(2, 1, 512, 512, device="cuda"); t = torch.rand(2, 1, 512, 512, device="cuda")
ys, gs = {}, {}
for math in ("fp32", "bf16x6", "f16x3"):
    m.set_math(math)
    with torch.no_grad():
        ys[math] = m(x).clone()
    tr = DataParallelTrainer(m, lr=0.0)
    tr.train_step(x, t)
    gs[math] = tr.grads.clone()
for math in ("bf16x6", "f16x3"):
    dy = float((ys[math] - ys["fp32"]).abs().max())
    dg = float((gs[math] - gs["fp32"]).abs().max() / gs["fp32"].abs().max())
    print(f"{math} vs fp32 mode: max |dy| {dy:.3e}   max |dgrad| / max |grad| {dg:.3e}   finite {bool(torch.isfinite(gs[math]).all())}")
