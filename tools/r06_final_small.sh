# round 6, after the balanced-grid rule: parity subset, the operating-point matrix, the 416 x 416 bench lines and kernel stats, the final
# before / after A/B (libxsd_hip_before.so = round-5 kernels).  bash tools/r06_final_small.sh   (GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O; cd $R
python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py tests/test_hip_abi_errors.py -m gpu -x -q > $O/pytest_subset.log 2>&1 || { tail -20 $O/pytest_subset.log; exit 1; }
tail -2 $O/pytest_subset.log
python3 tools/operating_points.py --seconds 1.5 --out $O/r06_operating_points.txt > $O/operating_points.log 2>&1 || exit 1
echo matrix done
python3 bench.py --batch 4 --tile 416 --steps 40 --warmup 5 --no-cpu-baseline > $O/r06_bench_dn_train_b4_416.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --batch 4 --tile 416 --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $O/r06_bench_sr_train_b4_416.json 2>/dev/null || exit 1
python3 bench.py --workload sr_fwd --batch 1 --tile 416 --steps 200 --warmup 10 --no-cpu-baseline --no-extra > $O/r06_bench_sr_fwd_b1_416.json 2>/dev/null || exit 1
python3 bench.py --batch 1 --tile 416 --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $O/r06_bench_dn_train_b1_416.json 2>/dev/null || exit 1
echo lines done
cd /tmp && export TMPDIR=/tmp
for C in "dn_train 4 416 20" "dn_fwd 1 416 100" "dn_train 1 416 60"; do
  set -- $C
  rm -rf $O/st_$1_b$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$1_b$2 -- python3 $R/bench.py --workload $1 --batch $2 --tile $3 --steps $4 --warmup 3 --no-profile --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/st_$1_b$2.json 2> $O/st_$1_b$2.err || exit 1
  cp $(ls $O/st_$1_b$2/*/*kernel_stats.csv | head -1) $O/r06_$1_b$2_$3_f16x3_kernel_stats.csv
done
echo stats done
cd $R
AB_BATCHES=1,4 AB_TILES=416,512 bash tools/ab_small.sh libxsd_hip_before.so product > $O/ab_small_final.txt 2>&1
echo ab done
