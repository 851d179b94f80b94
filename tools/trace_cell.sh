# kernel trace + stats of one operating-point cell: bash tools/trace_cell.sh <workload> <batch> <tile> <steps> <tag>   (GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --workload $1 --batch $2 --tile $3 --steps $4 --warmup 3 --no-profile --no-cpu-baseline --no-extra --no-sustained --no-psnr > $O/bench.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/raw/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv
rm -rf $O/raw
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-64s calls %6s avg %8.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel ms per step", tot / 1e6 / ($4 + 3))
PY
cat $O/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3))"
