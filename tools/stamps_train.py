"""Diagnostic (needs XSD_LIB=<diag library>): in-kernel phase stamps of the conv and weight-gradient kernels over one DN
train step.  usage: python tools/stamps_train.py <math> <batch>"""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + '/xmm-superres-denoise_amd', ROOT + '/tests/golden'): sys.path.insert(0, p)
from xmm_superres_denoise.models import GeneratorRRDB_DN
from xmm_superres_denoise.parallel import DataParallelTrainer
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
m = GeneratorRRDB_DN(1, 1, 32, 4).cuda().set_math(math)
x = torch.rand(B, 1, 512, 512, device='cuda'); t = torch.rand(B, 1, 512, 512, device='cuda')
tr = DataParallelTrainer(m)
tr.train_step(x, t); torch.cuda.synchronize()
eng = tr.engine
out = (ctypes.c_uint64 * 32)()
eng.L.xsd_debug_stamps(eng.h, 1, None)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); tr.train_step(x, t); t1.record(); torch.cuda.synchronize()
eng.L.xsd_debug_stamps(eng.h, 0, out)
v = list(out)
print(math, 'B', B, 'train step ms', t0.elapsed_time(t1))
names = ['prologue', 'prefetch issue', 'MFMA loop', 'epilogue', 'wait+barrier1', 'split+write+barrier2']
tot = sum(v[:6])
print(' conv: half-steps', v[6])
for n, c in zip(names, v[:6]): print(f'  {n:52s} {c / max(v[6], 1):10.0f} cycles/item  {100 * c / max(tot, 1):5.1f}%')
wn = ['staging rounds', 'MFMA walk', 'MFMA wave at the barrier', 'staging wave at the barrier', 'of the staging rounds: inside the counted data waits']
print(' wgrad: tiles', v[21])
for n, c in zip(wn, v[16:21]): print(f'  {n:52s} {c / max(v[21], 1):10.0f} cycles/tile')
if any(v[8:16]):
    print(' conv staging wave / youngest MFMA wave (cycles per half-step):')
    for n, c in zip(['input rounds', 'wait for DMA pieces', 'barrier', 'counted data waits', 'bookkeeping', 'youngest: MFMA loop', 'youngest: epilogue', 'youngest: barrier'], v[8:16]): print(f'  {n:52s} {c / max(v[6], 1):10.0f}')
if any(v[22:27]):
    ne = max(v[26], 1)
    print(' conv epilogue split, oldest MFMA wave (cycles per epilogue, %d epilogues):' % v[26])
    for n, c in zip(['entry -> first row operand / mask requests issued', 'first row arithmetic', 'second row (requests + arithmetic)', 'trailing vmcnt(0)'], v[22:26]): print(f'  {n:52s} {c / ne:10.0f}')
