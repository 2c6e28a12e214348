#!/bin/bash
# round 6: where the fused backward kernel (4 waves per SIMD, no spills) stops paying: step time against the Gaussian count
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6cross; mkdir -p $O; cd $R
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
for n in 20000 30000 40000 50000 65000 80000; do
  for leg in two fused; do
    export EG_BENCH_GAUSSIANS=$n
    if [ $leg = two ]; then export EG_TWO_KERNEL_BACKWARD=1; else unset EG_TWO_KERNEL_BACKWARD; fi
    for c in config1 config2; do
      timeout 300 python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N $n $leg $c (opacity ' + ('spread' if '$c' == 'config2' else 'init') + '): step us', round(1e3 * d['ms_per_step_median'], 2), 'M', int(d['config']['tile_intersections_M']))"
    done
  done
done
} 2>&1 | tee $O/summary.txt
