# Round profiles on the GPU box: rocprofv3 kernel stats of the default bench line and of the exact-fp32 mode, PMC traffic
# passes (separate --pmc runs, kernel-trace only), plain bench lines of every workload.  usage: bash tools/profile_round.sh r02
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in f16x3 bf16x6 fp32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$M -- python3 $R/bench.py --math $M --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/${TAG}_dn_train_b32_${M}_bench_under_rocprof.json 2> $O/stats_$M.err || exit 1
  cp $(ls $O/stats_$M/*/*kernel_stats.csv | head -1) $O/${TAG}_dn_train_b32_${M}_kernel_stats.csv
  echo "stats $M done"
done
cd $R
for M in f16x3 bf16x6; do bash tools/traffic.sh $M 32 > $O/traffic_$M.log 2>&1 || exit 1; cp gpurun_out/${TAG}_traffic_$M.json $O/; echo "traffic $M done"; done
python3 bench.py --steps 10 --warmup 3 > $O/${TAG}_bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --workload sr_fwd --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_sr_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload dn_fwd --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_dn_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --steps 4 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_sr_train.json 2>/dev/null || exit 1
python3 bench.py --loss paper --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $O/${TAG}_bench_dn_train_paper_loss.json 2>/dev/null || exit 1
python3 bench.py --input-pipeline --batch 16 --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $O/${TAG}_bench_dn_train_input_pipeline_b16.json 2>/dev/null || exit 1
XSD_DIST_BACKEND=gloo python3 bench.py --gpus 2 --batch 8 --steps 3 --warmup 1 --no-extra > $O/${TAG}_bench_dp2_gloo_one_gpu.json 2> $O/dp2.err || exit 1
echo all done
