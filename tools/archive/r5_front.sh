#!/bin/bash
# round 5: class boundary of the dispatch order under the XCD-aware placement (development build, EG_FRONT_SLICES)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5front; mkdir -p $O; cd $R
EG_DEV_SWITCHES=1 python -m edgegaussians_amd.build --force 2>&1 | tail -1
for fs in ${FS:-4 2 6 9 4}; do for c in config1 config2 config2i; do
  export EG_FRONT_SLICES=$fs
  a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py $a --steps 300 --warmup 20 --profile-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_front$fs.txt | grep "tile_sort\|composite_wave" | awk -v t="front=$fs $c" '{printf "%-18s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
  cd $R
done; done 2>&1 | tee $O/summary.txt
unset EG_FRONT_SLICES
python -m edgegaussians_amd.build --force 2>&1 | tail -1
