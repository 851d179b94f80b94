"""Speed of the other widths -- the plane kernels with 2-8 planes per tensor (64, 96, ... 256 filters, one image channel) and the
generic-width path (exact-fp32 kernels, csrc/generic_net.hip) -- next to the shipped 32-filter configuration:
DN forward and train step at 512x512.  Usage (through gpurun): python tools/generic_speed.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xmm-superres-denoise_amd"))
from xmm_superres_denoise.models import GeneratorRRDB_DN
from xmm_superres_denoise.parallel import DataParallelTrainer


def run(nf, cin, cout, blocks, B, math=None):
    torch.manual_seed(0)
    m = GeneratorRRDB_DN(cin, cout, nf, blocks).cuda()
    if math:
        m.set_math(math)
    x = torch.rand(B, cin, 512, 512, device="cuda")
    t = torch.rand(B, cout, 512, 512, device="cuda")
    with torch.no_grad():
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); fwd = time.perf_counter() - t0
    tr = DataParallelTrainer(m)
    tr.train_step(x, t); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.train_step(x, t); torch.cuda.synchronize(); st = time.perf_counter() - t0
    mac = 0
    def conv(ci, co): return 9 * ci * co
    rdb = sum(conv(nf * k, nf) for k in range(1, 6))
    mac = conv(cin, nf) + blocks * 3 * rdb + conv(nf, nf) + conv(nf, cout)
    fl = 2 * mac * 512 * 512 * B
    engine = ("planes" if nf % 32 == 0 else f"planes, padded to {(nf + 31) // 32 * 32}") if (cin <= 8 and cout <= 8 and nf <= 256) else "generic fp32"
    print(f"nf={nf} in={cin} out={cout} blocks={blocks} B={B} engine={engine} math={math or 'default'}: fwd {fwd*1e3:8.1f} ms = {fl/fwd/1e12:6.2f} TFLOP/s; "
          f"train step {st*1e3:8.1f} ms = {3*fl/st/1e12:6.2f} TFLOP/s ({B/st:.2f} tiles/s)", flush=True)


run(32, 1, 1, 4, 4, "fp32")
run(32, 1, 1, 4, 4, "f16x3")
run(32, 3, 3, 4, 4)
run(64, 1, 1, 4, 2)
run(8, 1, 1, 4, 8)
run(64, 1, 1, 4, 8)
run(128, 1, 1, 4, 2)
run(48, 1, 1, 4, 4)
run(16, 1, 1, 4, 8)
run(320, 1, 1, 1, 1)
