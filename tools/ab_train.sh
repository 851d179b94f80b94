# same-device A/B of the default train bench for the libraries in $LIBS (default: production vs lib/libxsd_hip_exp.so)
set -e
for L in ${LIBS:-libxsd_hip.so libxsd_hip_exp.so libxsd_hip.so}; do
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/abt_$L.log 2>&1 || { tail -20 gpurun_out/abt_$L.log; exit 1; }
  grep "^{" gpurun_out/abt_$L.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value'],2), round(d['ms_per_step'],1), 'conv', round(r['avg_launch_ms'],3), 'wgrad', round(r['wgrad_kernel']['avg_launch_ms'],3))"
done
