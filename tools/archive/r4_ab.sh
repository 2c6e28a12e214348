#!/bin/bash
# same-box compile-time A/B legs (rocprofv3 kernel stats, config 2 trained-like + initial opacity + config 4)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
run() { cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o r -- python $R/bench.py --config $2 $3 --steps 400 --warmup 20 --profile-only > /dev/null 2>&1; python $R/tools/rocpd_summary.py /tmp/ks_$1/r_results.db $O/kernel_stats_$1.txt | grep "composite_wave\|tile_sort\|footprint" | cut -c1-130; cd $R; }
leg() { echo "=== $1 [$2]"; EG_EXTRA_HIPCC_FLAGS="$2" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built"; run ${1}_c2 config2 ""; run ${1}_c4 config4 "--steps 150"; }
leg base ""
for l in "$@"; do leg "$(echo $l | tr -c 'A-Za-z0-9' '_')" "$l"; done
leg base2 ""
