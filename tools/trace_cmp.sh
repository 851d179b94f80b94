# per-dispatch comparison of the conv kernel between two library builds (same device): rocprofv3 kernel trace of one train step each
# usage: bash tools/trace_cmp.sh libA.so libB.so
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for L in $1 $2; do
  XSD_LIB=$R/xmm-superres-denoise_amd/lib/$L timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$L -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $R/gpurun_out/trace_$L.log 2>&1 || exit 1
done
python3 - "$R" "$1" "$2" <<'PY'
import csv, glob, sys
R, A, B = sys.argv[1:4]
def load(L):
    f = glob.glob(f"{R}/gpurun_out/trace_{L}/*/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv3x3_h2x" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
a, b = load(A), load(B)
n = len(a) // 3            # 3 steps traced (1 warm-up + 2): take the last
a, b = a[-n:], b[-n:]
print("conv launches per step", n, "sum ms", round(sum(a), 2), round(sum(b), 2))
# forward = first 61 launches (DN, 4 blocks), backward the rest
print("forward  ", round(sum(a[:61]), 2), round(sum(b[:61]), 2))
print("backward ", round(sum(a[61:]), 2), round(sum(b[61:]), 2))
for i in range(61, min(n, 61 + 12)): print(i, round(a[i], 3), round(b[i], 3))
PY
