# quick check of the default build: wgrad/conv kernel tests, then the DN train bench (B=32)
set -e
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "bf16x6" > gpurun_out/q_pytest.log 2>&1 || { tail -30 gpurun_out/q_pytest.log; exit 1; }
tail -1 gpurun_out/q_pytest.log
fmt='import sys,json
for d in map(json.loads, sys.stdin):
    r=d["roofline"]; print({k:d[k] for k in ("value","ms_per_step")}, "conv ms", round(r["avg_launch_ms"],3), "frac", round(r["frac"],3), "wgrad ms", r.get("wgrad_kernel",{}).get("avg_launch_ms"))'
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-extra > gpurun_out/q_train.log 2>&1
grep "^{" gpurun_out/q_train.log | python -c "$fmt"
timeout -k 10 300 python bench.py --workload dn_fwd --steps 6 --warmup 2 --no-extra > gpurun_out/q_fwd.log 2>&1
grep "^{" gpurun_out/q_fwd.log | python -c "$fmt"
