# gate experiment for pre-split ("H2") input planes: parity of the LDS-DMA staging variant, then forward bench with / without it
# (needs `make -C xmm-superres-denoise_amd/csrc exp EXPFLAGS=-DXSD_EXP_H2`)
set -e
export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_exp.so
XSD_H2=1 timeout -k 10 400 python -m pytest tests/test_hip_network.py -m gpu -x -q -k "f16x3" > gpurun_out/h2_tests.log 2>&1 || { tail -30 gpurun_out/h2_tests.log; exit 1; }
tail -2 gpurun_out/h2_tests.log
for H2 in 0 1 0 1; do
  XSD_H2=$H2 timeout -k 10 300 python bench.py --workload dn_fwd --steps 6 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/h2_fwd_$H2.log 2>&1 || { tail -20 gpurun_out/h2_fwd_$H2.log; exit 1; }
  grep "^{" gpurun_out/h2_fwd_$H2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('H2=$H2', round(d['value'],2), round(d['ms_per_step'],1), 'conv ms', round(r['avg_launch_ms'],3), 'launches', r['launches'])"
done
