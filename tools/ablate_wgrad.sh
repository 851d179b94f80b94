# timing experiments on diagnostic libraries: one DN train step (batch 8) under the weight-gradient kernel's ablation bits
# (1 no conversion, 2 no LDS writes; the conv kernel honours the same bits, so the train-step time moves with both -- read the
# wgrad stamp table, not the step time), one library per value (bash tools/build_abl.sh <values> first, on the CPU).
for A in ${@:-0}; do
  echo "== XSD_ABL=$A"
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_abl$A.so timeout -k 10 120 python tools/stamps_train.py f16x3 8 2>&1 | grep -A6 "wgrad:"
done
