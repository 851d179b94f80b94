# round 6: the operating-point matrix, then kernel traces of three small-grid cells (where does the time go?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/op6; mkdir -p $O
cd $R
python3 tools/operating_points.py --seconds 1.0 --out $O/r06_operating_points_before.txt > $O/op.log 2>&1 || { tail -20 $O/op.log; exit 1; }
echo "matrix done"
python3 bench.py --workload dn_train --batch 4 --tile 416 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/bench_dn_train_b4_416.json 2> $O/bench_b4.err || exit 1
cd /tmp && export TMPDIR=/tmp
for C in "dn_fwd 1 416 100" "dn_train 4 416 20" "dn_train 8 416 10" "sr_fwd 1 416 100"; do
  set -- $C
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$1_b$2_$3 -- python3 $R/bench.py --workload $1 --batch $2 --tile $3 --steps $4 --warmup 3 --no-profile --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/tr_$1_b$2_$3.json 2> $O/tr_$1_b$2_$3.err || exit 1
  cp $(ls $O/tr_$1_b$2_$3/*/*kernel_stats.csv | head -1) $O/r06_$1_b$2_$3_kernel_stats_before.csv
  echo "trace $1 b$2 $3 done"
done
