# What the 1400 W go to: package power / shader clock (sysfs, bench.py's PowerWatch) under (1) a device-to-device copy stream,
# (2) the conv's bare MFMA-wave stream in both shapes (tools/conv_shape_probe.hip, built on the CPU into tools/bin/), (3) the
# DN forward and (4) the DN train step.  usage: bash tools/power_table.sh  ->  gpurun_out/power_table.txt
O=gpurun_out/power_table.txt; : > $O
python - >> $O 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from bench import PowerWatch, device_pci
n = 1 << 30                                       # 4 GiB of fp32 read + 4 GiB written per copy
a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')
a.normal_()
for _ in range(5): b.copy_(a)
torch.cuda.synchronize()
with PowerWatch(0.05, pci=[device_pci(0)]) as pw:
    t0 = time.perf_counter()
    for _ in range(150): b.copy_(a)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("copy stream: %.2f TB/s (read + write)" % (150 * 2 * 4 * n / dt / 1e12), pw.summary())
PY
for S in 0 1 2; do echo "== conv MFMA-wave stream, shape $S (0: 32x32x16 shipped, 1: 16x16x32 as tap pairs with the half-empty ninth, 2: 16x16x32 zero-waste)" >> $O; python tools/power_of.py tools/bin/conv_shape_probe 400000 $S >> $O 2>&1; done
for W in dn_fwd dn_train; do python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$W', round(d['value'], 1), 'tiles/s', 'conv frac', round(r['frac'], 3), 'hbm GB/s (algorithmic, whole step)', round(r['whole_step']['algorithmic_GBps']), d.get('power'))" >> $O; done
cat $O
# (5) the train step WITHOUT its matrix instructions (experiment library: make -C xmm-superres-denoise_amd/csrc exp EXPFLAGS="-DX3_NOMFMA -DV3_NOMFMA"
#     EXP_OUT=../lib/libxsd_hip_nomfma.so; results are garbage): what staging, LDS, stores and HBM traffic cost by themselves
if [ -f xmm-superres-denoise_amd/lib/libxsd_hip_nomfma.so ]; then
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_nomfma.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('dn_train without MFMAs', round(d['value'], 1), 'tiles/s', round(d['ms_per_step'], 1), 'ms/step  conv', round(r['avg_launch_ms'], 3), 'ms  wgrad', round(r['wgrad_kernel']['avg_launch_ms'], 3), 'ms', d.get('power'))" >> $O
  tail -1 $O
fi
