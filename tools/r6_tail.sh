#!/bin/bash
# round 6: the three forms of the step's backward (eg_step_args.two_kernel_backward) -- bit-exactness, then kernel-trace averages
# and step time at configs 1 / 2: 1 = rounds 1-5 (footprint + 512-Gaussian workgroups), 2 = footprint + one-wave workgroups,
# 0 = default (one kernel up to 32768 Gaussians, else as 2)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6tail; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
stats() {
  for c in ${CONFIGS:-config1 config2}; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /tmp/ks_$c.log 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$1.txt > /dev/null
    echo "== $1 $c"; grep -E "gaussian_bwd_fused|gaussian_tail|footprint_bwd|project_bwd_emit|composite_wave_fwd|tile_sort" $O/kernel_stats_${c}_$1.txt | awk '{printf "   %-44s calls %6s avg %8s us\n", substr($1,1,44), $(NF-5), $(NF-3)}'
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
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "fused_backward_kernel or native_run or fused_train_step or data_parallel" 2>&1 | grep -v "$F" | tail -6
for m in 1 2 0; do EG_TWO_KERNEL_BACKWARD=$m stats mode$m; done
for extra in $EXTRA_LEGS; do
  EG_EXTRA_HIPCC_FLAGS="$extra" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  EG_TWO_KERNEL_BACKWARD=0 stats "mode0$(echo $extra | tr -d ' =-')"
done
} 2>&1 | tee $O/summary_${TAG:-run}.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
