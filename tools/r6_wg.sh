#!/bin/bash
# round 6: two workgroup-size legs -- footprint backward with 8 waves per workgroup (EG_FP_WAVES=8: footprints of 64 Gaussians sized
# once, as in the fused kernel) and the projection backward with 256 Gaussians per workgroup (EG_KPE=256)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6wg; mkdir -p $O; cd $R
stats() {
  for c in ${CONFIGS:-config1 config2 config3}; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /tmp/ks_$c.log 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$1.txt > /dev/null
    echo "== $1 $c"; grep -E "gaussian_bwd_fused|footprint_bwd|project_bwd_emit" $O/kernel_stats_${c}_$1.txt | awk '{printf "   %-44s calls %6s avg %8s us\n", substr($1,1,44), $(NF-5), $(NF-3)}'
    cd $R
  done
}
{
for leg in "" "-DEG_FP_WAVES=8" "-DEG_KPE=256" "" "-DEG_FP_WAVES=8" "-DEG_KPE=256"; do
  EG_EXTRA_HIPCC_FLAGS="$leg" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  EG_TWO_KERNEL_BACKWARD=1 stats "base$(echo $leg | tr -d ' =-')"
done
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
