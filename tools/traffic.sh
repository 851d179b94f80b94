# HBM traffic of the MFMA kernels from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE, kernel-trace only.  usage: bash tools/traffic.sh <math> [batch] [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=$1; B=${2:-32}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_${M}_$C      # (a second run in one call must not find the first run's files)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_${M}_$C -- python3 $R/bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-profile --no-extra --no-sustained --no-psnr --math $M > $R/gpurun_out/traffic_${M}_$C.log 2>&1 || exit 1
done
python3 $R/tools/traffic_read.py $M $B ${3:-r05}
