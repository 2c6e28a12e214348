#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o r -- python $R/bench.py --config config2 --steps 500 --warmup 50 --profile-only > /dev/null 2>&1
python $R/tools/kernel_per_view.py /tmp/pv/r_results.db 50 | tee $O/kernel_per_view_config2.txt
