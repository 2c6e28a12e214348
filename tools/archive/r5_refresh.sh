#!/bin/bash
# round 5: refresh of the headline evidence on the final tree (default bench line with same-run traffic, kernel stats at configs 1-4, the GPU suite)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ev; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
rm -f $R/gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" | tail -15 > $O/pytest_gpu.log
cp $R/gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
( time timeout 900 python bench.py ) 2>$O/bench_default.err | tail -1 > $O/bench_default.json
timeout 600 python bench.py --config config1 --no-extra 2>/dev/null | tail -1 > $O/bench_config1.json
cd /tmp && export TMPDIR=/tmp
for c in config1 config2 config2i config3 config4; do
  rm -rf /tmp/ev_$c
  a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ev_$c -o r -- python $R/bench.py $a --steps 300 --warmup 20 --profile-only > /dev/null 2>$O/prof_$c.err
  python $R/tools/rocpd_summary.py /tmp/ev_$c/r_results.db $O/kernel_stats_$c.txt > /dev/null
  python $R/tools/timeline_gaps.py /tmp/ev_$c/r_results.db > $O/timeline_gaps_$c.txt
done
cd $R
tail -3 $O/pytest_gpu.log; head -c 500 $O/bench_default.json; echo; for c in config1 config2 config3 config4; do head -6 $O/kernel_stats_$c.txt | cut -c1-120; done
