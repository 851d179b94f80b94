# builds (here, on the CPU: hipcc cross-compiles) one diagnostic library per ablation value for tools/ablate_conv.sh /
# ablate_wgrad.sh / convfree_probe.sh: lib/libxsd_hip_abl<N>.so = -DXSD_DIAG -DXSD_ABL=<N>.  The kernels' ablation bits are
# compile-time constants (a run-time value puts the hand-counted loads under branches hipcc cannot keep exact).
# usage: bash tools/build_abl.sh 0 1 2 3 4 7 16 19
set -e
for A in "$@"; do
  make -s -C xmm-superres-denoise_amd/csrc -j8 exp EXPFLAGS="-DXSD_DIAG -DXSD_ABL=$A" EXP_OUT=../lib/libxsd_hip_abl$A.so 2>&1 | grep -i "error" && exit 1
  echo "built libxsd_hip_abl$A.so"
done
