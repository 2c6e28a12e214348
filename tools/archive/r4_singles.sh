#!/bin/bash
# same-box A/B of the sort kernel with / without the single-slice-tile class sums (compile-time)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
run() { cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o r -- python $R/bench.py --config $2 $3 --steps 400 --warmup 20 --profile-only > /dev/null 2>&1; python $R/tools/rocpd_summary.py /tmp/ks_$1/r_results.db $O/kernel_stats_$1.txt | head -6; cd $R; }
echo "=== default build"; run singles_c2 config2 ""; run singles_c2i config2 "--init-opacity"
EG_EXTRA_HIPCC_FLAGS="-DEG_NO_SINGLES" python -m edgegaussians_amd.build --force 2>&1 | tail -1
echo "=== -DEG_NO_SINGLES (EG_SINGLES_LAST off by construction: cbef = ctot = 0)"; run nosingles_c2 config2 ""; run nosingles_c2i config2 "--init-opacity"
echo "=== default build again"; python -m edgegaussians_amd.build --force 2>&1 | tail -1; run singles2_c2 config2 ""
