# f16x3 kernels: kernel parity, then the train bench next to bf16x6 (stops at the first failure)
set -e
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "f16x3" > gpurun_out/h2_k.log 2>&1 || { tail -30 gpurun_out/h2_k.log; exit 1; }
tail -1 gpurun_out/h2_k.log
for M in f16x3 bf16x6; do
timeout -k 10 300 python bench.py --math $M --steps 6 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/h2_train_$M.log 2>&1 || { tail -30 gpurun_out/h2_train_$M.log; exit 1; }
grep "^{" gpurun_out/h2_train_$M.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$M', round(d['value'],2), round(d['ms_per_step'],1), 'conv', round(r['avg_launch_ms'],3), 'wgrad', round(r['wgrad_kernel']['avg_launch_ms'],3))"
done
