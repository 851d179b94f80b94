# round 6: same-device A/B of library builds at the reference's small operating points (alternating, twice)
#   bash tools/ab_small.sh libA.so libB.so ...    (paths relative to xmm-superres-denoise_amd/lib/; "product" = the default library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_small; mkdir -p $O
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = product ]; then unset XSD_LIB; else export XSD_LIB=$R/xmm-superres-denoise_amd/lib/$L; fi
    python3 $R/tools/operating_points.py --seconds 1.0 --batches ${AB_BATCHES:-1,4} --tiles ${AB_TILES:-416,512} --maths ${AB_MATHS:-f16x3} ${AB_ONLY:+--only $AB_ONLY} 2>/dev/null | grep -v "^# json" | sed "s/^/[$rep $L] /"
  done
done
