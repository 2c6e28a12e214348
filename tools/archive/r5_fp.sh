#!/bin/bash
# same-box A/B of the footprint backward's two walks (round 5): rocprofv3 kernel stats of bench.py --profile-only at
# configs 2 / 3 / 4 with the row walk never / always / at thresholds.   gpurun --timeout 1500 -- 'bash tools/r5_fp.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5fp; mkdir -p $O; cd $R
CFGS=${CFGS:-"config2 config3 config4"}
THRS=${THRS:-"2147483647"}  # (the row walk and its threshold are gone: the script now only times the kernel)
run() {  # tag config thr
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o r -- python $R/bench.py --config $2 --steps ${STEPS:-200} --warmup 20 --profile-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_$1/r_results.db $O/kernel_stats_$1.txt | grep "footprint_bwd\|composite_wave" | awk -v t="$2 rows_min=$3" '{printf "%-34s %-28s avg %8s us  min %8s  max %8s\n", t, substr($1,1,28), $4, $5, $6}'
  cd $R
}
for c in $CFGS; do for t in $THRS; do run ${c}_$t $c $t; done; done 2>&1 | tee $O/summary.txt
