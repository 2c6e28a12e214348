#!/bin/bash
# round 6: XCD-aware record placement above 2048 tiles (EG_FLAG_XCD_PREFIX) -- tests, then configs 3 / 4 / abc800 with the
# placement on and off (development build: EG_XCD_LARGE=0 keeps the dense two-class records of round 5): forward time (kernel trace)
# and fabric traffic (PMC passes of bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6xcd; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
{
EG_DEV_SWITCHES=1 python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "dispatch_order or large_grid or grid_shapes or fused_backward_kernel or native_run or item_overflow or config3_size or overflow" 2>&1 | grep -v "$F" | tail -8
for c in ${CONFIGS:-config3 config4 abc800}; do
  for leg in 1 0; do
    export EG_XCD_LARGE=$leg
    echo "== $c EG_XCD_LARGE=$leg"
    timeout 900 python bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('   step us', round(1e3 * d['ms_per_step_median'], 2), 'M', int(d['config']['tile_intersections_M']))
        print('   kernel trace avg us', {k: round(v, 2) for k, v in (d.get('kernel_trace_avg_us') or {}).items()})
        t = d.get('traffic_bytes_per_step_by_stage') or {}
        print('   traffic MB per step', {k: round(v / 1e6, 1) for k, v in t.items()})
        rf = d['roofline']; print('   roofline', rf['kernel'], 'alg MB', round(rf['algorithmic_bytes_per_launch'] / 1e6, 1), 'frac', round(rf['frac'], 4), 'traffic MB', None if rf.get('traffic') is None else round(rf['traffic'] / 1e6, 1))"
  done
done
} 2>&1 | tee $O/summary_${TAG:-run}.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
