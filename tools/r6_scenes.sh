#!/bin/bash
# round 6: S scenes per GPU (config 1) with the one-kernel backward (default) and with the two kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6scenes; mkdir -p $O; cd $R
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "fused_backward_kernel or multi_scene" 2>&1 | tail -3
for rep in 1 2; do for m in 0 1; do for S in 1 2 4 8; do
  EG_TWO_KERNEL_BACKWARD=$m timeout 300 python bench.py --config config1 --scenes-per-gpu $S 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('two_kernel_backward $m S $S:', round(d['value'] / 1e6), 'M G*v/s', round(d['aggregate_us_per_scene_step'], 2), 'us per scene-step')"
done; done; done
} 2>&1 | tee $O/summary.txt
