# timing experiments on the diagnostic library: phase stamps of the bf16x6 conv under diagnostic knobs (XSD_ABLATE bits)
# role-split kernel (default): 1 no split, 2 no input LDS writes, 4 no input loads / counted waits, 16 empty input descriptors
# bit 20 (1048576) selects the unified-wave kernel instead; its bits: 1 no input conversion/LDS writes, 16 no input loads,
# 2 no weight path, 4 no epilogue, 8 no MFMAs.  Results are garbage under any of these bits: timing only.
export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so
for A in ${@:-0}; do
  echo "== XSD_ABLATE=$A"
  XSD_ABLATE=$A timeout -k 10 120 python tools/stamps.py bf16x6 8 2>&1 | grep -v amdgpu.ids
done
