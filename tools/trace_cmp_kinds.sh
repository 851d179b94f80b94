# per-launch-kind comparison of the conv kernel between two library builds (same device, one gpurun call): rocprofv3 kernel trace of
# one train step each, launches grouped by K-loop length (the grouping of tools/trace_by_kind.sh).  usage: bash tools/trace_cmp_kinds.sh libA.so libB.so
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for L in $1 $2; do
  rm -rf $R/gpurun_out/trace_$L
  XSD_LIB=$R/xmm-superres-denoise_amd/lib/$L timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$L -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $R/gpurun_out/trace_$L.log 2>&1 || exit 1
done
python3 - "$R" "$1" "$2" <<'PY'
import csv, glob, sys
R, A, B = sys.argv[1:4]
def load(L):
    f = glob.glob(f"{R}/gpurun_out/trace_{L}/*/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv3x3_h2x" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    n = len(d) // 3
    return d[-n:]
a, b = load(A), load(B)
print(f"{'kind':34s} {A:>22s} {B:>22s}   delta")
def row(name, ia):
    va, vb = sum(a[i] for i in ia) / len(ia), sum(b[i] for i in ia) / len(ia)
    print(f"{name:34s} {va:19.3f} ms {vb:19.3f} ms  {100 * (vb / va - 1):+6.1f} %")
for c in range(5): row(f"forward conv{c+1} ({c+1} plane steps)", [k * 5 + c for k in range(12)])
row("forward trunk (1 plane step)", [60])
row("backward trunk^T (1 plane step)", [61])
for k in range(5): row(f"backward dS_{4-k} ({k+1} plane steps)", [62 + q * 5 + k for q in range(12)])
print(f"{'all conv launches of a step':34s} {sum(a):19.2f} ms {sum(b):19.2f} ms  {100 * (sum(b) / sum(a) - 1):+6.1f} %")
PY
