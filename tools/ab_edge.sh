# round 6: edge kernels before / after (roofline.edge of the bench line: algorithmic bytes / HIP-event time), same device, alternating
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for L in libxsd_hip_before.so product; do
    if [ "$L" = product ]; then unset XSD_LIB; else export XSD_LIB=$R/xmm-superres-denoise_amd/lib/$L; fi
    for W in "dn_train 32" "sr_train 16" "dn_train 4 --tile 416"; do
      set -- $W
      python3 $R/bench.py --workload $1 --batch $2 $3 $4 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-sustained --no-psnr 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d['roofline']['edge']
print('[$rep $L] $1 b$2 $3$4: %.2f tiles/s (unprofiled %.2f);' % (d['value'], d['unprofiled']['value']), ' '.join('%s %.0f GB/s %.0f us' % (k, v['achieved'], v['avg_launch_us']) for k, v in e.items() if isinstance(v, dict) and k.startswith('edge')), '; edge+elementwise share %.4f' % e['share_of_profiled_kernel_time'])"
    done
  done
done
