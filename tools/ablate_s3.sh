# timing experiments on the diagnostic library: phase stamps of the bf16x6 conv under diagnostic knobs
# (bits: 1 no input conversion/LDS writes, 16 no input loads, 2 no weight path, 4 no epilogue, 8 no MFMAs; bits 8-12: stagger
# of waves 4-7 in s_sleep units, 0 = the built-in default)
export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_diag.so
for A in ${@:-0}; do
  echo "== XSD_ABLATE=$A"
  XSD_ABLATE=$A timeout -k 10 120 python tools/stamps.py bf16x6 8 2>&1 | grep -v amdgpu.ids
done
