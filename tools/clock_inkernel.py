import sys, os, ctypes, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+'/xmm-superres-denoise_amd', ROOT+'/tests/golden'): sys.path.insert(0,p)
os.environ.setdefault('XSD_P16','v2')
from xmm_superres_denoise.models import GeneratorRRDB_DN
torch.manual_seed(0)
m = GeneratorRRDB_DN(1,1,32,4).cuda().set_math('bf16x3_p16')
x = torch.rand(8,1,512,512,device='cuda')
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    eng=m._engine; out=(ctypes.c_uint64*16)()
    eng.L.xsd_debug_stamps(eng.h,1,None)
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    eng.L.xsd_debug_stamps(eng.h,0,out)
v=list(out)
print('ablate',os.environ.get('XSD_ABLATE','0'),'half-steps',v[6],'MFMA loop: cycles/half-step',v[0]/v[6],'wall ns/half-step',v[1]*10/v[6],'clock GHz in MFMA loop',v[0]/(v[1]*10),'| kernel clock GHz',v[2]/(v[3]*10))
