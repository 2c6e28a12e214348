#!/bin/bash
# round 6: the fused per-Gaussian backward kernel (backward_fused.hip) against the two-kernel path -- bit-exactness test,
# then per-kernel averages (rocprofv3 kernel trace) and step time at configs 1 / 2 for several occupancy targets
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6fused; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
CONFIGS=${CONFIGS:-"config1 config2"}
stats() {  # $1 leg name, env in front
  for c in $CONFIGS; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /tmp/ks_$c.log 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$1.txt > /dev/null
    echo "== $1 $c"; grep -E "gaussian_bwd_fused|footprint_bwd|project_bwd_emit|composite_wave_fwd|tile_sort" $O/kernel_stats_${c}_$1.txt | awk '{printf "   %-44s calls %6s avg %8s us\n", substr($1,1,44), $(NF-5), $(NF-3)}'
    cd $R
    timeout 300 python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step us', round(1e3 * d['ms_per_step'], 2), 'windows', [round(1e3 * x, 2) for x in d.get('ms_per_step_windows', [])])"
  done
}
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "fused_backward_kernel or native_run or fused_train_step" 2>&1 | grep -v "$F" | tail -5
EG_TWO_KERNEL_BACKWARD=1 stats two_kernel
stats fused_w8
for w in $WAVES; do
  EG_EXTRA_HIPCC_FLAGS="-DEG_BF_WAVES_PER_EU=$w" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  stats fused_w$w
done
} 2>&1 | tee $O/summary_${TAG:-run}.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
