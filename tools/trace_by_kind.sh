# per-dispatch durations of the conv kernel over one DN train step (batch 32, f16x3), grouped by the launch's position in the
# plan: forward conv_c of the dense blocks (c input planes), backward input-gradient launches (5 - j gradient planes), trunk.
# usage (through gpurun): bash tools/trace_by_kind.sh
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_kind -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $R/gpurun_out/trace_kind.log 2>&1 || exit 1
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/trace_kind/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "conv3x3_h2x" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = len(d) // 3
d = d[-n:]                       # the last of the 3 traced steps
print("conv launches per step", n, "sum ms", round(sum(d), 2))
fwd, bwd = d[:61], d[61:]
# forward: 12 dense blocks x conv1..5 (1..5 input planes), then the trunk conv
for c in range(5):
    v = [fwd[b * 5 + c] for b in range(12)]
    print(f"forward conv{c+1} ({c+1} plane steps): {sum(v)/len(v):.3f} ms per launch = {sum(v)/len(v)/(c+1):.3f} ms per plane step")
print(f"forward trunk (1 plane step): {fwd[60]:.3f} ms")
# backward: first launch = trunk input-gradient; then per dense block (reverse) 5 input-gradient launches with 1..5 gradient planes
print(f"backward trunk^T (1 plane step): {bwd[0]:.3f} ms")
for k in range(5):
    v = [bwd[1 + b * 5 + k] for b in range(12)]
    print(f"backward dS_{4-k} ({k+1} plane steps): {sum(v)/len(v):.3f} ms per launch = {sum(v)/len(v)/(k+1):.3f} ms per plane step")
PY
