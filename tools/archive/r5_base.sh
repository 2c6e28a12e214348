#!/bin/bash
# Round-5 baseline on the GPU box: GPU tests, the default bench line, kernel stats at configs 1-4.
#   gpurun --timeout 1800 -- 'bash tools/r5_base.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5base; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
if [ "${TESTS:-1}" = "1" ]; then
  rm -f $R/gpurun_out/parity_report.jsonl
  timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" | tail -25 > $O/pytest_gpu.log
  cp $R/gpurun_out/parity_report.jsonl $O/ 2>/dev/null
fi
[ "${BENCH:-1}" = "1" ] && ( timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json )
cd /tmp; export TMPDIR=/tmp
for c in ${CFGS:-config1 config2 config3 config4}; do
  rm -rf /tmp/ks_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o r -- python $R/bench.py --config $c --steps ${STEPS:-300} --warmup 20 --profile-only > /dev/null 2>$O/prof_$c.err
  python $R/tools/rocpd_summary.py /tmp/ks_$c/r_results.db $O/kernel_stats_$c.txt > /dev/null
  echo "== $c"; head -8 $O/kernel_stats_$c.txt | cut -c1-150
done
cd $R; tail -3 $O/pytest_gpu.log; head -c 600 $O/bench_default.json; echo
