# The energy budget of the strict mode (bf16x6), as tools/power_table.sh does it for f16x3: (1) the bf16 MFMA-wave stream alone
# (xsd_probe_mfma_stream, 6 s, watts and clock from sysfs), (2) the DN train step, (3) the train step WITHOUT its matrix
# instructions (experiment library: make -C xmm-superres-denoise_amd/csrc exp EXPFLAGS="-DXSD_EXP_NOMFMA -DV3S_NOMFMA"
# EXP_OUT=../lib/libxsd_hip_nomfma_strict.so; results are garbage).  usage: bash tools/power_table_strict.sh -> gpurun_out/power_table_strict.txt
O=gpurun_out/power_table_strict.txt; : > $O
python - >> $O 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'xmm-superres-denoise_amd')
from bench import PowerWatch, device_pci
from xmm_superres_denoise.engine.engine import Engine
eng = Engine("dn", 1, 1, 32, 1)
for fmt in ("bf16", "f16"):
    eng.probe_mfma_stream(fmt, 1.0)
    pw = PowerWatch(0.05, pci=[device_pci(0)]); pw.start()
    r = eng.probe_mfma_stream(fmt, 6.0)
    pw.stop()
    print("bare %s MFMA-wave stream: %.0f dense TFLOP/s at %.0f MHz in-kernel" % (fmt, r["mfma_tflops"], 1e3 * r["sclk_ghz"]), pw.summary())
PY
fmt='
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print(TAG, round(d["value"], 1), "tiles/s", round(d["ms_per_step"], 1), "ms/step  conv", round(r["avg_launch_ms"], 3), "ms x", r["launches"] // d["steps"], " wgrad", round(r["wgrad_kernel"]["avg_launch_ms"], 3), "ms x", r["wgrad_kernel"]["launches"] // d["steps"], d.get("power"))'
python bench.py --math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-psnr --no-sustained 2>/dev/null | python -c "TAG='dn_train bf16x6'$fmt" >> $O
if [ -f xmm-superres-denoise_amd/lib/libxsd_hip_nomfma_strict.so ]; then
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_nomfma_strict.so python bench.py --math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-psnr --no-sustained 2>/dev/null | python -c "TAG='dn_train bf16x6 without MFMAs'$fmt" >> $O
fi
cat $O
