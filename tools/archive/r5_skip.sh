#!/bin/bash
# round 5: empty tiles skipped (runtime switch of a development build) -- kernel times + step time per leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5skip; mkdir -p $O; cd $R
EG_DEV_SWITCHES=1 python -m edgegaussians_amd.build --force 2>&1 | tail -1
for sk in 1 0 1 0; do for c in config1 config2; do
  export EG_SKIP_EMPTY=$sk
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}_skip$sk.txt | grep "tile_sort\|composite_wave" | awk -v t="skip=$sk $c" '{printf "%-16s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
  cd $R
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('skip=$sk $c: %.2f us/step (windows %s)' % (d['ms_per_step']*1e3, [round(x*1e3,2) for x in d['ms_per_step_windows']]))"
done; done 2>&1 | tee $O/summary.txt
unset EG_SKIP_EMPTY
python -m edgegaussians_amd.build --force 2>&1 | tail -1
