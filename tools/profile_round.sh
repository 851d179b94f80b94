# Round profiles on the GPU box: rocprofv3 kernel stats of the default bench line (f16x3) and of the strict / exact modes, PMC
# traffic passes (separate --pmc runs, kernel-trace only), SQ counter passes of the f16x3 train step, plain bench lines of every
# workload.  Everything lands in gpurun_out/prof_<tag>/ named for profiles/.  usage: bash tools/profile_round.sh r06 [part]   (part 1: rocprof + PMC passes; part 2: bench lines, operating points, probes; no part: both)
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}; PART=${2:-12}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
if [[ $PART == *1* ]]; then
cd /tmp && export TMPDIR=/tmp
for M in f16x3 bf16x6 fp32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$M -- python3 $R/bench.py --math $M --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_dn_train_b32_${M}_bench_under_rocprof.json 2> $O/stats_$M.err || exit 1
  cp $(ls $O/stats_$M/*/*kernel_stats.csv | head -1) $O/${TAG}_dn_train_b32_${M}_kernel_stats.csv
  echo "stats $M done"
done
cd $R
for M in f16x3 bf16x6; do bash tools/traffic.sh $M 32 $TAG > $O/traffic_$M.log 2>&1 || exit 1; cp gpurun_out/${TAG}_traffic_$M.json $O/; echo "traffic $M done"; done
# SQ counters of the f16x3 kernels over one train step at the bench batch (matrix pipe busy, issue stalls, LDS conflicts)
PMC_BATCH=32 PMC_WORKLOAD=dn_train bash tools/pmc.sh f16x3 > $O/pmc_f16x3.log 2>&1 || exit 1
python3 tools/pmc_read.py f16x3 > $O/${TAG}_pmc_sq_f16x3_dn_train_b32.txt || exit 1
# ... and of the strict mode's kernels (round 5: the weight gradient's (pixel half, row pair) walk; are its LDS accesses conflict-free?)
PMC_BATCH=32 PMC_WORKLOAD=dn_train bash tools/pmc.sh bf16x6 > $O/pmc_bf16x6.log 2>&1 || exit 1
python3 tools/pmc_read.py bf16x6 > $O/${TAG}_pmc_sq_bf16x6_dn_train_b32.txt || exit 1
echo "sq counters done"
# round 6: the SR side (configs[1] / configs[3]'s per-GPU share): kernel stats of the 1024^2 kernels
cd /tmp
for W in sr_fwd sr_train; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py --workload $W --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_${W}_b16_f16x3_bench_under_rocprof.json 2> $O/stats_$W.err || exit 1
  cp $(ls $O/stats_$W/*/*kernel_stats.csv | head -1) $O/${TAG}_${W}_b16_f16x3_kernel_stats.csv
  echo "stats $W done"
done
# ... and the reference's own operating point: DN train, batch 4, 416 x 416
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b4_416 -- python3 $R/bench.py --batch 4 --tile 416 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_dn_train_b4_416_f16x3_bench_under_rocprof.json 2> $O/stats_b4_416.err || exit 1
cp $(ls $O/stats_b4_416/*/*kernel_stats.csv | head -1) $O/${TAG}_dn_train_b4_416_f16x3_kernel_stats.csv
cd $R
fi
if [[ $PART == *2* ]]; then
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/bench_default.err || exit 1
# the by-construction-fp32 mode as a first-class line of its own (headline slot, own cpu_baseline; its PMC traffic file is above)
python3 bench.py --math bf16x6 --no-extra --steps 20 --warmup 5 > $O/${TAG}_bench_bf16x6.json 2> $O/bench_bf16x6.err || exit 1
python3 bench.py --workload sr_fwd --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_sr_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload dn_fwd --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_dn_fwd.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --steps 4 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_sr_train.json 2>/dev/null || exit 1
python3 bench.py --loss paper --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_bench_dn_train_paper_loss.json 2>/dev/null || exit 1
python3 bench.py --input-pipeline --batch 16 --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_bench_dn_train_input_pipeline_b16.json 2>/dev/null || exit 1
python3 bench.py --batch 16 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/${TAG}_bench_dn_train_b16.json 2>/dev/null || exit 1
# the RCCL code path on the one GPU: a one-rank nccl process group with the reduce path forced on (parallel.collectives_on)
XSD_FORCE_DP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-sustained --no-psnr 2> $O/dp1_rccl.err | grep '^{' > $O/${TAG}_bench_dp1_rccl_one_gpu.json || exit 1      # (RCCL / gloo print banners on stdout: the .json holds the line only)
XSD_DIST_BACKEND=gloo python3 bench.py --gpus 2 --batch 8 --steps 3 --warmup 1 --no-extra 2> $O/dp2.err | grep '^{' > $O/${TAG}_bench_dp2_gloo_one_gpu.json || exit 1
# round 6: the reference's own operating points (batch 1 / 4 / 8, 416 x 416): lines with roofline / edge blocks, then the whole matrix
python3 bench.py --batch 4 --tile 416 --steps 40 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_dn_train_b4_416.json 2>/dev/null || exit 1
python3 bench.py --workload sr_train --batch 4 --tile 416 --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $O/${TAG}_bench_sr_train_b4_416.json 2>/dev/null || exit 1
python3 bench.py --workload sr_fwd --batch 1 --tile 416 --steps 200 --warmup 10 --no-cpu-baseline --no-extra > $O/${TAG}_bench_sr_fwd_b1_416.json 2>/dev/null || exit 1
python3 tools/operating_points.py --seconds 1.5 --out $O/${TAG}_operating_points.txt > $O/operating_points.log 2>&1 || exit 1
./tools/bin/launch_floor_probe > $O/${TAG}_launch_floor.txt 2>&1 || exit 1
for c in "1 416" "1 512" "4 416" "32 512"; do XSD_LIB=$R/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so python3 tools/stamps_small.py f16x3 $c >> $O/${TAG}_small_grid_stamps.txt 2>/dev/null || exit 1; done
echo all done
fi
