# tools/convfree_probe.py under rocprofv3, both modes, same device: average durations of the conv and weight-gradient kernels
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
# (bash tools/build_abl.sh 0 1 first, on the CPU)
for M in 0 1 0 1; do
  rm -rf $R/gpurun_out/cfp_$M
  XSD_LIB=$R/xmm-superres-denoise_amd/lib/libxsd_hip_abl$M.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfp_$M -- python3 $R/tools/convfree_probe.py $M > $R/gpurun_out/cfp_$M.log 2>&1 || { tail -5 $R/gpurun_out/cfp_$M.log; exit 1; }
  grep "^mode" $R/gpurun_out/cfp_$M.log
  python3 - "$R/gpurun_out/cfp_$M" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "conv3x3_h2x" in r["Name"] or "wgrad_h2x" in r["Name"]:
        print("   %-28s calls %3s  avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
