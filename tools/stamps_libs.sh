# wgrad stamp table (cycles per tile, DN train step) for a list of library builds under lib/: bash tools/stamps_libs.sh <batch> <lib> ...
B=${1:-8}; shift
for L in "$@"; do
  echo "== $L"
  XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L timeout -k 10 150 python tools/stamps_train.py f16x3 $B 2>&1 | grep -A5 "wgrad:"
done
