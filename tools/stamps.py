import sys, os, ctypes, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+'/xmm-superres-denoise_amd', ROOT+'/tests/golden'): sys.path.insert(0,p)
from xmm_superres_denoise.models import GeneratorRRDB_DN
math = sys.argv[1] if len(sys.argv)>1 else 'f16x3'
B = int(sys.argv[2]) if len(sys.argv)>2 else 8
torch.manual_seed(0)
m = GeneratorRRDB_DN(1,1,32,4).cuda().set_math(math)
x = torch.rand(B,1,512,512,device='cuda')
with torch.no_grad():
    m(x); torch.cuda.synchronize()
    eng = m._engine
    out = (ctypes.c_uint64*32)()
    eng.L.xsd_debug_stamps(eng.h, 1, None)
    t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True)
    t0.record(); m(x); t1.record(); torch.cuda.synchronize()
    eng.L.xsd_debug_stamps(eng.h, 0, out)
v=list(out)
names=['prologue','prefetch issue','MFMA loop','epilogue','wait+barrier1','split+write+barrier2']
tot=sum(v[:6])
print(math,'B',B,'fwd ms',t0.elapsed_time(t1),'items',v[6])
for n,c in zip(names,v[:6]): print(f'  {n:24s} {c/max(v[6],1):10.0f} cycles/item  {100*c/tot:5.1f}%')
if v[7]: print(f'  in-kernel clock (stamped cycles / s_memrealtime at 100 MHz): {tot/v[7]*0.1:.3f} GHz')
if any(v[8:13]):
    print('  staging wave 0 (role-split kernel):')
    for n,c in zip(['descriptors, weight DMA issue, input rounds','wait for the DMA pieces','barrier','of the rounds: inside the counted data waits','cursor + tile offsets of the next half-steps'],v[8:13]): print(f'  {n:24s} {c/max(v[6],1):10.0f} cycles/item')
if any(v[13:16]):
    print('  youngest MFMA wave (role-split kernel):')
    for n,c in zip(['MFMA loop','epilogue','wait+barrier1'],v[13:16]): print(f'  {n:24s} {c/max(v[6],1):10.0f} cycles/item')
