#!/bin/bash
# round 5: the sort kernel's prefix waves (512-thread variant) -- phase profile + kernel times per leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5sort2; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for pw in 4 8 2; do
  EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF -DEG_SORT_PREF_WAVES_512=$pw" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  timeout 300 python tools/sort_prof.py config2 --spread 2>&1 | grep -v "$F" > $O/sort_phases_config2_pw$pw.txt
  echo "== prefix waves $pw (profile build)"; grep -v "^    " $O/sort_phases_config2_pw$pw.txt | head -14
  EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PREF_WAVES_512=$pw" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_pw$pw
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_pw$pw -o r -- python $R/bench.py --config config2 --steps 300 --warmup 20 --profile-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_pw$pw/r_results.db $O/kernel_stats_pw$pw.txt | grep "tile_sort\|composite_wave" | awk -v t="pw$pw config2" '{printf "%-16s %-34s avg %s us\n", t, substr($0,1,34), $(NF-3)}'
  cd $R
done 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
