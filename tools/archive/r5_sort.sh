#!/bin/bash
# same-box compile-time A/B of the tile sort kernel (round 5): buckets per thread (EG_SORT_BM), keys ranked side by side
# (EG_SORT_RANK_G), 256- against 512-thread workgroups (EG_SORT_WIDE, development build).
#   gpurun --timeout 1800 -- 'bash tools/r5_sort.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5sort; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
CFGS=${CFGS:-"config1 config2"}
run() {  # tag
  for c in $CFGS; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1_$c -o r -- python $R/bench.py --config $c --steps ${STEPS:-200} --warmup 20 --profile-only > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$1_$c/r_results.db $O/kernel_stats_$1_$c.txt | grep "tile_sort\|composite_wave" | awk -v t="$1 $c" '{printf "%-24s %-30s avg %s us\n", t, substr($0,1,30), $(NF-3)}'
    cd $R
  done
}
leg() {  # tag flags wide
  EG_DEV_SWITCHES=1 EG_EXTRA_HIPCC_FLAGS="$2" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -3
  if [ -n "$3" ]; then export EG_SORT_WIDE=$3; else unset EG_SORT_WIDE; fi
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "binning_bit_exact or prefix_switch" 2>&1 | grep -v "$F" | tail -1
  run $1
}
{
if [ -n "$LEGS" ]; then
  IFS=';' read -ra L <<< "$LEGS"
  for l in "${L[@]}"; do tag=${l%%=*}; fl=${l#*=}; leg "$tag" "$fl" ""; done
else
leg base "" ""
leg bm2 "-DEG_SORT_BM=2" ""
leg waves6 "-DEG_SORT_WAVES=6" ""
leg bm2waves6 "-DEG_SORT_BM=2 -DEG_SORT_WAVES=6" ""
leg base2 "" ""
fi
} 2>&1 | tee $O/summary.txt
unset EG_SORT_WIDE
python -m edgegaussians_amd.build --force 2>&1 | tail -1
