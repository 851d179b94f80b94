"""Where a SMALL-GRID conv launch spends its time (round 6): phase stamps of the f16x3 conv kernel (diagnostic library,
`make -C xmm-superres-denoise_amd/csrc diag`, XSD_LIB=.../libxsd_hip_diag.so) over one DN forward at batch B and tile T, turned into
microseconds PER WORKGROUP AND LAUNCH next to the launch's wall time from HIP events -- the difference is what a launch costs
outside its waves' lifetime (dispatch, kernel-argument fetch, end-of-kernel cache write-back).
usage: XSD_LIB=... python3 tools/stamps_small.py [math] [B] [T]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/xmm-superres-denoise_amd', ROOT + '/tests/golden'):
    sys.path.insert(0, p)
from xmm_superres_denoise.models import GeneratorRRDB_DN  # noqa: E402

math = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 416
torch.manual_seed(0)
m = GeneratorRRDB_DN(1, 1, 32, 4).cuda().set_math(math)
x = torch.rand(B, 1, T, T, device='cuda')
NCONV = 61
tiles = B * ((T + 15) // 16) * ((T + 31) // 32)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nwg = min(tiles, ncu)
if os.environ.get("XSD_EXP_GRID"):      # an experiment build with a forced grid (-DXSD_GRID_ENV): that many workgroups, no plan-time tuning
    nwg = min(tiles, int(os.environ["XSD_EXP_GRID"]))
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    eng = m._engine
    if os.environ.get("XSD_EXP_GRID") and hasattr(eng, "set_grid_tuning"):
        eng.set_grid_tuning(False)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
    if not os.environ.get("XSD_EXP_GRID") and hasattr(eng.L, "xsd_debug_persistent_grid"):      # the grid the launcher really uses (csrc/xsd_kernels.h: persistent_grid)
        eng.L.xsd_debug_persistent_grid.argtypes = [ctypes.c_int, ctypes.c_int]
        nwg = int(eng.L.xsd_debug_persistent_grid(tiles, ncu))
    out = (ctypes.c_uint64 * 32)()
    eng.L.xsd_debug_stamps(eng.h, 1, None)
    eng.profile_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); m(x); t1.record(); torch.cuda.synchronize()
    prof = eng.profile_read(0)
    eng.profile_enable(False)
    eng.L.xsd_debug_stamps(eng.h, 0, out)
v = list(out)
names = ['prologue (kernel entry -> first barrier)', 'prefetch issue', 'MFMA loop', 'epilogue', 'wait+barrier']
tot = sum(v[:5])
ghz = tot / v[7] * 0.1 if v[7] else float('nan')
launches = prof['launches']
print(f"{math} B {B} tile {T}: fwd {t0.elapsed_time(t1):.3f} ms; {launches} conv launches, event time {1e3 * prof['ms'] / launches:.1f} us per launch; "
      f"{tiles} tiles on {nwg} workgroups; in-kernel clock {ghz:.3f} GHz")
per = launches * nwg
print(f"  per workgroup and launch (oldest MFMA wave), mean over {per} workgroup runs:")
for n, c in zip(names, v[:5]):
    print(f"    {n:44s} {c / per / ghz / 1e3:8.2f} us   ({c / per:9.0f} cycles)")
print(f"    {'kernel body, entry -> exit (s_memrealtime)':44s} {v[7] / per * 0.01:8.2f} us")
print(f"    {'half-steps per workgroup and launch':44s} {v[6] / per:8.2f}")
print(f"  launch wall time (events) - mean body = {1e3 * prof['ms'] / launches - v[7] / per * 0.01:.2f} us (dispatch + argument fetch + tail of the slowest workgroup + end-of-kernel write-back)")
