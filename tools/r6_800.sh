#!/bin/bash
# round 6: kPrefixHereMaxTiles 2048 -> 2560 (the reference's native 800 x 800 = 2500 tiles on the small-grid path): suite + abc800 + configs 1 / 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6800; mkdir -p $O; cd $R
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
timeout 1800 python -m pytest tests -m gpu -q -x --tb=short ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
for c in abc800 config1 config2; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps 300 --warmup 20 --profile-only > /tmp/ks_$c.log 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_${c}.txt > /dev/null
  echo "== $c"; head -8 $O/kernel_stats_${c}.txt | cut -c1-130
  cd $R
  timeout 300 python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   step us', [round(1e3 * x, 2) for x in d['ms_per_step_windows']], 'M', int(d['config']['tile_intersections_M']))"
done
} 2>&1 | tee $O/summary.txt
