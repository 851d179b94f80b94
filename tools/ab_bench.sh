# same-device A/B of library builds on the DEFAULT bench line (batch 32 x 512 x 512 DN train; f16x3 + the bf16x6 extra leg), alternating twice
#   bash tools/ab_bench.sh libA.so libB.so ...   ("product" = the default library)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = product ]; then unset XSD_LIB; else export XSD_LIB=$R/xmm-superres-denoise_amd/lib/$L; fi
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-sustained ${AB_BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; e = d.get('extra')
print('[$rep $L] %.2f tiles/s (unprofiled %.2f) conv %.4f ms wgrad %.3f ms edge_expand %.0f us W %.0f MHz %.0f' % (d['value'], d['unprofiled']['value'], r['avg_launch_ms'], r['wgrad_kernel']['avg_launch_ms'], r['edge']['edge_expand']['avg_launch_us'], d['power']['avg_w'], d['power']['sclk_mhz']), ('| bf16x6 %.2f tiles/s conv %.4f ms' % (e['value'], e['roofline']['avg_launch_ms'])) if e else '')"
  done
done
