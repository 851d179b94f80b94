# effective clock of the conv kernels: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration  (MI355X_MICROARCH.md "DVFS give-back")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 7 8; do
  XSD_ABLATE=$a timeout -k 10 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$a -- python3 $R/bench.py --steps 1 --warmup 1 --batch 8 --no-cpu-baseline --no-profile --math bf16x3_p16 --workload dn_fwd > $R/gpurun_out/clk_$a.log 2>&1 || exit 1
done
echo ok
