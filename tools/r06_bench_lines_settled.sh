#!/bin/bash
# The secondary bench lines of the round-6 profile set, retaken with timed regions of >= 3 s after >= 1.5 s of warm-up: the first set
# (tools/profile_round.sh: 4-6 steps after 1-2 warm-ups) timed 0.3-1.1 s that began before the package had settled at its 1400 W cap
# (their own `power` blocks: 1177-1385 W average where the 20-step headline reads 1399 W), which flatters a line by up to ~4 %.
# usage (GPU box, from the repo root):  bash tools/r06_bench_lines_settled.sh
O=gpurun_out/r06_settled
mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-sustained --no-psnr"
python3 bench.py --workload sr_fwd --steps 80 --warmup 30 --no-cpu-baseline > $O/r06_bench_sr_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload dn_fwd --steps 40 --warmup 15 --no-cpu-baseline > $O/r06_bench_dn_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --steps 12 --warmup 5 --no-cpu-baseline > $O/r06_bench_sr_train.json 2>/dev/null || exit 1
echo "lines 1-3 done"
python3 bench.py --loss paper --steps 12 --warmup 5 $Q > $O/r06_bench_dn_train_paper_loss.json 2>/dev/null || exit 1
python3 bench.py --batch 16 --steps 30 --warmup 10 $Q > $O/r06_bench_dn_train_b16.json 2>/dev/null || exit 1
XSD_FORCE_DP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python3 bench.py --gpus 1 --steps 12 --warmup 5 $Q 2> $O/dp1_rccl.err | grep '^{' > $O/r06_bench_dp1_rccl_one_gpu.json || exit 1
XSD_DIST_BACKEND=gloo python3 bench.py --gpus 2 --batch 8 --steps 24 --warmup 10 --no-extra --no-cpu-baseline 2> $O/dp2.err | grep '^{' > $O/r06_bench_dp2_gloo_one_gpu.json || exit 1
echo "lines 4-7 done"
python3 bench.py --batch 4 --tile 416 --steps 150 --warmup 60 --no-extra > $O/r06_bench_dn_train_b4_416.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --batch 4 --tile 416 --steps 150 --warmup 60 --no-cpu-baseline --no-extra > $O/r06_bench_sr_train_b4_416.json 2>/dev/null || exit 1
echo "lines 8-9 done"
python3 bench.py --workload sr_fwd --batch 1 --tile 416 --steps 1200 --warmup 400 --no-cpu-baseline --no-extra > $O/r06_bench_sr_fwd_b1_416.json 2>/dev/null || exit 1
python3 bench.py --batch 1 --tile 416 --steps 500 --warmup 200 --no-extra > $O/r06_bench_dn_train_b1_416.json 2>/dev/null || exit 1
python3 bench.py --workload dn_fwd --batch 1 --tile 416 --steps 1500 --warmup 500 --no-extra > $O/r06_bench_dn_fwd_b1_416.json 2>/dev/null || exit 1
echo "lines 10-12 done"
for f in $O/r06_bench_*.json; do python3 - "$f" <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d.get("power") or {}
print("%-52s value %8.2f unprofiled %8.2f  timed %5.2f s  %s W  sclk %s" % (sys.argv[1].split("/")[-1], d["value"], (d.get("unprofiled") or {}).get("value", float("nan")),
      d["steps"] * d["ms_per_step"] / 1e3, p.get("avg_w"), p.get("sclk_mhz")))
EOF
done
