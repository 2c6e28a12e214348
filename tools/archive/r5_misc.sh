#!/bin/bash
# round 5: leftover runtime switches of a development build under the final record placement
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5misc; mkdir -p $O; cd $R
EG_DEV_SWITCHES=1 python -m edgegaussians_amd.build --force 2>&1 | tail -1
leg() {  # tag env...
  tag=$1; shift
  for c in config1 config2; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
    env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_$tag.txt | grep "tile_sort\|composite_wave" | awk -v t="$tag $c" '{printf "%-22s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
    cd $R
  done
}
{
leg default EG_NOP=1
leg sort_plain_order EG_SORT_MIDDLE_OUT=0
leg xcd_blocks_4x4 EG_XCD_SHIFT=2
leg default2 EG_NOP=1
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
