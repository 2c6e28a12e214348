#!/bin/bash
# same-box compile-time A/B of the forward's head / look-back loads (round 5)
#   gpurun --timeout 1800 -- 'bash tools/r5_fwd.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5fwd; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
CFGS=${CFGS:-"config1 config2 config2i config4"}
run() {  # tag
  for c in $CFGS; do
    a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1_$c -o r -- python $R/bench.py $a --steps ${STEPS:-300} --warmup 20 --profile-only > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$1_$c/r_results.db $O/kernel_stats_$1_$c.txt | grep "tile_sort\|composite_wave\|project_bwd_emit\|footprint" | awk -v t="$1 $c" '{printf "%-24s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
    cd $R
  done
}
leg() {  # tag flags
  EG_EXTRA_HIPCC_FLAGS="$2" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -3
  [ "${TESTS:-1}" = "1" ] && timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_floats.py -q -x -k "chained or stop or dispatch_order or speculat or oracle_floats or binning" 2>&1 | grep -v "$F" | tail -1
  run $1
}
{
if [ -n "$LEGS" ]; then
  IFS=';' read -ra L <<< "$LEGS"
  for l in "${L[@]}"; do tag=${l%%=*}; fl=${l#*=}; leg "$tag" "$fl"; done
else
leg new ""
leg old "-DEG_AB_OLD_HEAD -DEG_AB_OLD_ANCHOR -DEG_AB_OLD_LOOK"
leg oldhead "-DEG_AB_OLD_HEAD"
leg oldanchor "-DEG_AB_OLD_ANCHOR"
leg new2 ""
fi
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
