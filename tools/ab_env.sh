# same-library, same-device A/B of an engine switch read from the environment: bash tools/ab_env.sh VAR   (alternates VAR=0 / VAR=1)
# (round 6: only the test-hooks variant of the library reads these switches -- make -C xmm-superres-denoise_amd/csrc hooks)
export XSD_LIB=${XSD_LIB:-$GRAFT_REPO_ROOT/xmm-superres-denoise_amd/lib/libxsd_hip_hooks.so}
V=$1
for R in 0 1 0 1; do
  export $V=$R
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; w = r['wgrad_kernel']
print('$V=$R', round(d['value'], 2), 'tiles/s', round(d['ms_per_step'], 1), 'ms  conv', round(r['avg_launch_ms'], 4), 'x', r['launches'], ' wgrad', round(w['avg_launch_ms'], 4), 'x', w['launches'], '= %.1f ms/step' % (w['avg_launch_ms'] * w['launches'] / d['steps']), 'W', d['power']['avg_w'], 'MHz', d['power']['sclk_mhz'])"
done
