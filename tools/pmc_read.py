import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for i in (1, 2):
    f = glob.glob(os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_{i}', '*', '*counter_collection.csv'))[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3x3' not in k and not (('wgrad_h2x' in k or 'wgrad_s3x' in k or 'wgrad_mfma' in k) and 'reduce' not in k): continue   # (not the edge layers' edge_wgrad_*)
        name = 'conv' if 'conv3x3' in k else 'wgrad'
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_WAIT_INST_LDS'): cnt[(name, i)] += 1
for name, d in agg.items():
    print('==', name, 'dispatches', cnt[(name, 1)])
    wc = d['SQ_WAVE_CYCLES']
    for k, v in sorted(d.items()):
        print(f'  {k:28s} {v:.4g}  ({v / wc:.3f} of WAVE_CYCLES)')
    if d.get('SQ_INSTS_VMEM_RD'): print('  avg VMEM latency-ish (LEVEL/INSTS):', d['SQ_INST_LEVEL_VMEM'] / d['SQ_INSTS_VMEM_RD'])
    if d.get('SQ_INSTS_LDS'): print('  avg LDS level/inst:', d['SQ_INST_LEVEL_LDS'] / d['SQ_INSTS_LDS'])
