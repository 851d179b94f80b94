# Same-device A/B of library builds in one math mode: bash tools/ab_libs_math.sh MATH lib1.so lib2.so ...  (alternating, twice)
# one line per run: tiles/s, conv / weight-gradient kernel averages, watts and clock (the pool's devices differ by up to 9 %:
# only numbers of ONE call are compared)
M=$1; shift
for R in 1 2; do
for L in "$@"; do
  export XSD_LIB=$PWD/xmm-superres-denoise_amd/lib/$L
  timeout -k 10 300 python bench.py --math $M --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-psnr --sustained-seconds 0.5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; w = r['wgrad_kernel']; p = d.get('power') or {}
print('$M $L', round(d['value'], 2), 'tiles/s', round(d['ms_per_step'], 1), 'ms  conv', round(r['avg_launch_ms'], 4), ' wgrad', round(w['avg_launch_ms'], 4), 'x', w['launches'], '= %.1f ms/step' % (w['avg_launch_ms'] * w['launches'] / d['steps']), ' ', p.get('avg_w'), 'W', p.get('sclk_mhz'), 'MHz')" || exit 1
done
done
