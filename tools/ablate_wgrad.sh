# timing experiments on the diagnostic library: one DN train step (batch 8) under the weight-gradient kernel's ablation bits
# (1 no conversion / LDS writes, 16 no loads, 8 no MFMAs); the conv kernel honours the same bits, so the train-step time
# moves with both -- read the wgrad stamp table, not the step time.
export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so
for A in ${@:-0}; do
  echo "== XSD_ABLATE=$A"
  XSD_ABLATE=$A timeout -k 10 120 python tools/stamps_train.py bf16x6 8 2>&1 | grep -A6 "wgrad:"
done
