#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ev_roctx
timeout 300 rocprofv3 --kernel-trace --marker-trace --stats -d /tmp/ev_roctx -o r -- python $R/bench.py --config config2 --roctx --steps 200 --warmup 10 --profile-only > /dev/null 2>$O/prof_roctx.err
python $R/tools/roctx_summary.py /tmp/ev_roctx/r_results.db $O/roctx_ranges_config2.txt
