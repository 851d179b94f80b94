# timing experiments on the diagnostic library: the DN forward pass (batch 32, f16x3) under the conv kernel's staging ablation
# bits (conv3x3_h2x.hip: 1 no split, 2 no input LDS writes, 4 no input loads / no weight DMA / no counted waits, 16 empty input
# descriptors = loads issued, no memory traffic).  Results are garbage under any bit: read the stamp tables only.
export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so
for A in ${@:-0}; do
  echo "== XSD_ABLATE=$A"
  XSD_ABLATE=$A timeout -k 10 150 python tools/stamps.py f16x3 32 2>&1 | grep -v "^$" || exit 1
done
