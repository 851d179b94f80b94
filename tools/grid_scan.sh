# round 6: the persistent conv grid scanned directly -- DN forward at batch $1 and tile $2 with the grid forced to each of $3.. workgroups
# (experiment build -DXSD_GRID_ENV: lib/libxsd_hip_gridenv.so).  bash tools/grid_scan.sh <batch> <tile> <G> <G> ...
R=$GRAFT_REPO_ROOT; B=$1; T=$2; shift 2
export XSD_LIB=$R/xmm-superres-denoise_amd/lib/libxsd_hip_gridenv.so
for G in "$@"; do
  XSD_EXP_GRID=$G python3 $R/tools/operating_points.py --seconds 0.7 --only dn_fwd --batches $B --tiles $T --maths f16x3 2>/dev/null | grep "^dn_fwd" | awk -v g=$G '{printf "batch %d tile %d grid %3d: %8.2f tiles/s %8.3f ms\n", $3, $4, g, $5, $6}'
done
