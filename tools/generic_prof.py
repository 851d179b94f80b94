"""One generic-width train step under rocprofv3 (kernel split of csrc/generic_net.hip): python tools/generic_prof.py <nf> <in> <out> <B>"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xmm-superres-denoise_amd"))
from xmm_superres_denoise.models import GeneratorRRDB_DN
from xmm_superres_denoise.parallel import DataParallelTrainer
nf, cin, cout, B = (int(a) for a in sys.argv[1:5])
torch.manual_seed(0)
m = GeneratorRRDB_DN(cin, cout, nf, 4).cuda()
x = torch.rand(B, cin, 512, 512, device="cuda"); t = torch.rand(B, cout, 512, 512, device="cuda")
tr = DataParallelTrainer(m)
for _ in range(2):
    tr.train_step(x, t)
torch.cuda.synchronize()
