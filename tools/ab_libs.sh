# A/B of library builds on ONE device: the libraries named in $LIBS (e.g. LIBS="libxsd_hip.so libxsd_hip_exp.so libxsd_hip.so"
# after `make exp EXPFLAGS=...`)
set -e
fmt='import sys,json
for d in map(json.loads, sys.stdin):
    r=d["roofline"]; print({k:d[k] for k in ("value","ms_per_step")}, "conv ms", round(r["avg_launch_ms"],3), "frac", round(r["frac"],3), "wgrad ms", r.get("wgrad_kernel",{}).get("avg_launch_ms"))'
for L in ${LIBS:-libxsd_hip.so libxsd_hip_exp.so libxsd_hip.so}; do
  echo "== $L"
  export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L
  timeout -k 10 300 python bench.py --workload dn_fwd --steps 6 --warmup 2 --no-extra > gpurun_out/ab_$L.fwd.log 2>&1
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-extra > gpurun_out/ab_$L.train.log 2>&1
  cat gpurun_out/ab_$L.fwd.log gpurun_out/ab_$L.train.log | grep "^{" | python -c "$fmt"
done
