#!/bin/bash
# round 6: the first timed window of a process that starts right behind another GPU job -- bench lines behind a rocprofv3 pass and behind
# a pytest run, three each
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6fw; mkdir -p $O; cd $R
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 windows us', [round(1e3 * x, 2) for x in d['ms_per_step_windows']])"; }
{
for c in config2 config1 config2; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fw
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fw -o r -- python $R/bench.py --config $c --steps 100 --warmup 10 --profile-only > /dev/null 2>&1
  cd $R
  python bench.py --config $c --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | line "behind rocprofv3 $c"
done
for i in 1 2; do
  python -m pytest tests/test_gpu_parity.py -q -x -k "fused_backward_kernel or binning" > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | line "behind pytest config2"
done
} 2>&1 | tee $O/summary.txt
