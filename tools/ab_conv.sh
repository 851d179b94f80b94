# Same-device A/B of conv-kernel variants: for every library in $LIBS (files under xmm-superres-denoise_amd/lib/; the production
# library first AND last, so that a drift of the device over the call shows) the f16x3 kernel parity tests, then the DN
# forward bench (conv launches only) and the DN train bench; one line per run.  Usage (through gpurun, from the repo root):
#   LIBS="libxsd_hip.so libxsd_hip_expA.so libxsd_hip.so" bash tools/ab_conv.sh
set -e
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; w=r.get("wgrad_kernel",{})
print("%-22s %-9s %8.2f tiles/s %8.2f ms/step  conv %.4f ms (frac %.4f)  wgrad %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], ("%.4f ms" % w["avg_launch_ms"]) if w else "-"))'
for L in ${LIBS:-libxsd_hip.so libxsd_hip_exp.so libxsd_hip.so}; do
  export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L
  timeout -k 10 200 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "f16x3 and (forward or backward)" > gpurun_out/abc_$L.pytest.log 2>&1 || { echo "$L: parity FAILED"; tail -20 gpurun_out/abc_$L.pytest.log; continue; }
  if [ -n "$ABC_NET" ]; then timeout -k 10 300 python -m pytest tests/test_hip_network.py -x -q -m gpu -k "f16x3 and (golden or fresh)" > gpurun_out/abc_$L.net.log 2>&1 || { echo "$L: network parity FAILED"; tail -20 gpurun_out/abc_$L.net.log; continue; }; fi
  timeout -k 10 200 python bench.py --workload dn_fwd --steps ${STEPS:-8} --warmup 3 --no-extra --no-cpu-baseline 2> gpurun_out/abc_$L.fwd.err | grep "^{" | python -c "$fmt" $L fwd
  timeout -k 10 200 python bench.py --steps ${STEPS:-8} --warmup 3 --no-extra --no-cpu-baseline 2> gpurun_out/abc_$L.train.err | grep "^{" | python -c "$fmt" $L train
done
