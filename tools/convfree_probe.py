"""What conversion-free staging would buy the f16x3 conv (input-gradient) and weight-gradient kernels, on realistic operands.

The diagnostic library's ablation bit 1 makes the staging waves write a loaded float4 to LDS as it is (no fp32 -> 2 x fp16
split).  On fp32 planes that produces degenerate fp16 images (NaN), and a chip that is power-limited under these kernels then
holds 2.2 GHz instead of 1.66 -- useless as a timing.  Here the planes are FILLED with pre-split data (per float4:
four fp16 high terms | four fp16 low terms of normal variates), so the raw copy leaves realistic two-term images in LDS: the
kernels run the shipped instruction stream minus the conversions, on operands that toggle like the real ones.  Results are not
checked (the scales come from the reinterpreted bits); only the kernel durations are read, by rocprofv3:
    XSD_LIB=.../libxsd_hip_abl<0|1>.so rocprofv3 --kernel-trace --stats ... -- python3 tools/convfree_probe.py <0|1>
argument 0: shipped staging on fp32 normal variates; 1: raw staging on the pre-split fill."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "xmm-superres-denoise_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
mode = int(sys.argv[1])
# the library decides: libxsd_hip_abl1.so (raw staging) or libxsd_hip_abl0.so (tools/convfree_probe.sh)
from xmm_superres_denoise.engine import Engine
from xmm_superres_denoise.engine._lib import check
from util_hip import ptr_array

B, H, W, n_in = 16, 512, 512, 3
gen = torch.Generator(device="cuda").manual_seed(5)


def plane():
    x = torch.randn((B, H, W, 32), device="cuda", generator=gen)
    if not mode:
        return x
    h = x.half()
    l = ((x - h.float()) * 2048.0).half()
    q = torch.cat([h.view(B, H, W, 8, 4), l.view(B, H, W, 8, 4)], dim=-1).contiguous()     # [.., quad, 4 hi | 4 lo] = 16 B
    return q.view(torch.float32).view(B, H, W, 32)


e = Engine("dn", 1, 1, 32, 1)
e.set_math("f16x3")
xs = [plane() for _ in range(n_in)]
g = plane()
w = torch.randn((32, 32 * n_in, 3, 3), device="cuda", generator=gen) / (288 * n_in) ** 0.5
dxs = [torch.empty((B, H, W, 32), device="cuda") for _ in range(n_in)]
dw, db = torch.empty_like(w), torch.empty((32,), device="cuda")
for _ in range(6):
    check(e.L.xsd_test_conv3x3_bwd(e.h, ptr_array(xs), n_in, w.data_ptr(), g.data_ptr(), ptr_array(dxs), dw.data_ptr(), db.data_ptr(), B, H, W, None))
torch.cuda.synchronize()
print("mode", mode, "finite dw:", bool(torch.isfinite(dw).all()), "finite dx:", bool(torch.isfinite(dxs[0]).all()))
