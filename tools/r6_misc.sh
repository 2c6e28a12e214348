#!/bin/bash
# round 6: full GPU suite on the current tree, then two A/Bs -- the fused backward's early loads (EG_BF_HOIST) and the tile sort
# with one workgroup per TWO tiles (EG_SORT_GRID_DIV=2: middle-out rank b paired with rank b + T / 2)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6misc; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
stats() {  # $1 leg, $2 kernel pattern
  for c in ${CONFIGS:-config1 config2}; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /tmp/ks_$c.log 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$1.txt > /dev/null
    echo "== $1 $c"; grep -E "$2" $O/kernel_stats_${c}_$1.txt | awk '{printf "   %-44s calls %6s avg %8s us\n", substr($1,1,44), $(NF-5), $(NF-3)}'
    cd $R
    timeout 300 python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step us median', round(1e3 * d['ms_per_step_median'], 2), 'windows', [round(1e3 * x, 2) for x in d.get('ms_per_step_windows', [])])"
  done
}
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -15; cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null; fi
CONFIGS=config1 stats hoist1 "gaussian_bwd_fused"
EG_EXTRA_HIPCC_FLAGS="-DEG_BF_HOIST=0" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
CONFIGS=config1 stats hoist0 "gaussian_bwd_fused"
EG_EXTRA_HIPCC_FLAGS="" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
CONFIGS=config1 stats hoist1b "gaussian_bwd_fused"
EG_DEV_SWITCHES=1 python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "binning_bit_exact or dispatch_order or native_run" 2>&1 | grep -v "$F" | tail -3
for d in 1 2 1 2; do EG_SORT_GRID_DIV=$d stats sortdiv$d "tile_sort|composite_wave_fwd"; done
EG_SORT_GRID_DIV=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "binning_bit_exact or dispatch_order or native_run or fused_train_step" 2>&1 | grep -v "$F" | tail -3
} 2>&1 | tee $O/summary_${TAG:-run}.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
