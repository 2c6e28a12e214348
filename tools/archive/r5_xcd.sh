#!/bin/bash
# round 5: XCD-aware record placement -- tests, kernel times and the forward's fabric traffic per placement (compile-time legs)
#   gpurun --timeout 2400 -- 'bash tools/r5_xcd.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5xcd; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
CFGS=${CFGS:-"config1 config2 config2i"}
run() {  # tag
  for c in $CFGS; do
    a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$1_$c
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$1_$c -o r -- python $R/bench.py $a --steps ${STEPS:-300} --warmup 20 --profile-only > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ks_$1_$c/r_results.db $O/kernel_stats_$1_$c.txt | grep "tile_sort\|composite_wave" | awk -v t="$1 $c" '{printf "%-16s %-34s calls %5s avg %s us\n", t, substr($0,1,34), $(NF-5), $(NF-3)}'
    cd $R
  done
  if [ "${TRAFFIC:-1}" = "1" ]; then
    timeout 600 python bench.py --config config2 --no-cpu-baseline --no-extra --steps 300 2>/dev/null | tail -1 > $O/bench_config2_$1.json
    python -c "
import json; d=json.loads(open('$O/bench_config2_$1.json').read()); r=d['roofline']
print('$1 config2: %.2f us/step, forward %.2f us, traffic %.1f MB = %.2fx algorithmic (%.1f MB), by stage %s' % (d['ms_per_step']*1e3, r['avg_launch_us'], (r['traffic'] or 0)/1e6, (r['traffic'] or 0)/r['algorithmic_bytes_per_launch'], r['algorithmic_bytes_per_launch']/1e6, {k: round(v/1e6,1) for k,v in (d.get('traffic_bytes_per_step_by_stage') or {}).items()}))"
  fi
}
leg() {  # tag flags tests
  EG_EXTRA_HIPCC_FLAGS="$2" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -3
  if [ "$3" = "1" ]; then
    timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" > $O/pytest_$1.log; tail -${TAIL:-25} $O/pytest_$1.log | cut -c1-220
  fi
  run $1
}
{
if [ -n "$LEGS" ]; then
  IFS=';' read -ra L <<< "$LEGS"
  for l in "${L[@]}"; do tag=${l%%=*}; fl=${l#*=}; leg "$tag" "$fl" 0; done
else
leg xcd1 "" 1
leg xcd0 "-DEG_XCD_SHIFT_DEFAULT=0" 0
leg xcd1b "" 0
fi
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
