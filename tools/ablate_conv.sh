# timing experiments on diagnostic libraries: the DN forward pass (batch 32, f16x3) under the conv kernel's staging ablation
# bits (conv3x3_h2x.hip: 1 no split, 2 no input LDS writes, 4 no input loads / no weight DMA / no counted waits, 16 empty input
# descriptors = loads issued, no memory traffic), one library per value (bash tools/build_abl.sh <values> first, on the CPU).
# Results are garbage under any bit, and every bit that makes the operand images degenerate raises the clock: read the CYCLES.
for A in ${@:-0}; do
  echo "== XSD_ABL=$A"
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/libxsd_hip_abl$A.so timeout -k 10 150 python tools/stamps.py f16x3 32 2>&1 | grep -v "^$\|amdgpu.ids" || exit 1
done
