#!/bin/bash
# round 5: quick check of a build -- binning / record / forward tests, kernel times at configs 1 / 2 / 2i (+ 3 / 4 with CFGS), sort phases
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5check; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_floats.py -q -x --tb=short -k "${KEXPR:-binning or prefix_switch or dispatch_order or chained or stop or overflow or fused_step_small or multi_scene or batched}" 2>&1 | grep -v "$F" | tail -${TAIL:-4} | cut -c1-200
for c in ${CFGS:-config1 config2 config2i}; do
  a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py $a --steps ${STEPS:-300} --warmup 20 --profile-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_$c.txt | grep "tile_sort\|composite_wave\|footprint\|project_bwd_emit" | awk -v t="$c" '{printf "%-10s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
  cd $R
done
if [ "${SORTPROF:-1}" = "1" ]; then
  EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  for c in config2 config1; do timeout 300 python tools/sort_prof.py $c --spread 2>&1 | grep -v "$F" > $O/sort_phases_$c.txt; grep -v "^    " $O/sort_phases_$c.txt | head -13; done
  python -m edgegaussians_amd.build --force 2>&1 | tail -1
fi
