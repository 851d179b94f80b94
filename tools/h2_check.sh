# f16x3 conv: kernel parity, phase stamps, forward bench (stops at the first failure)
set -e
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "f16x3" > gpurun_out/h2_k.log 2>&1 || { tail -30 gpurun_out/h2_k.log; exit 1; }
tail -1 gpurun_out/h2_k.log
XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so timeout -k 10 200 python tools/stamps.py f16x3 32 > gpurun_out/h2_stamps.log 2>&1 || { tail -30 gpurun_out/h2_stamps.log; exit 1; }
grep -v amdgpu.ids gpurun_out/h2_stamps.log
for M in f16x3 bf16x6; do
timeout -k 10 300 python bench.py --math $M --workload dn_fwd --steps 6 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/h2_fwd_$M.log 2>&1 || { tail -30 gpurun_out/h2_fwd_$M.log; exit 1; }
grep "^{" gpurun_out/h2_fwd_$M.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
