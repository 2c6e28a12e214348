#!/bin/bash
# round 5: footprint backward with the workgroup's footprints sized once (LDS) -- compile-time A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5fps; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
leg() {
  EG_EXTRA_HIPCC_FLAGS="$2" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  [ "$3" = "1" ] && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_floats.py tests/test_gpu_fullsize.py -q -x --tb=short -k "backward or fused or fullsize or oracle_floats or big_footprints or batched" 2>&1 | grep -v "$F" | tail -2
  for c in config1 config2 config3 config4; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$1.txt | grep "footprint" | awk -v t="$1 $c" '{printf "%-16s %-24s avg %s us\n", t, substr($0,1,24), $(NF-3)}'
    cd $R
  done
}
{
leg shared "-DEG_FP_SHARED_WALK=1" 1
leg base "" 0
leg shared2 "-DEG_FP_SHARED_WALK=1" 0
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
