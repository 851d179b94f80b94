for K in 1 2 4 1 2 4; do
  export XSD_WGRAD_SPLIT=$K
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; w = r['wgrad_kernel']
print('split $K', round(d['value'], 2), 'tiles/s  conv', round(r['avg_launch_ms'], 4), ' wgrad', round(w['avg_launch_ms'], 4), 'x', w['launches'], '= %.1f ms/step' % (w['avg_launch_ms'] * w['launches'] / d['steps']))"
done
rd() { python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], {k: (d[k]['launches'], round(d[k]['traffic_bytes_per_launch'] * d[k]['launches'] / 2e9, 1)) for k in ('conv', 'wgrad')}, 'GB per step')" "$1"; }
for K in 1 4; do XSD_WGRAD_SPLIT=$K bash tools/traffic.sh f16x3 32 sp$K | rd "split $K traffic"; done
