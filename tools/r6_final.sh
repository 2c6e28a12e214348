#!/bin/bash
# round 6 evidence on the final tree: the GPU suite + parity report, the default bench line (the driver's command), bench lines of
# the sibling configurations, rocprofv3 kernel stats + launch gaps + SQ counters per configuration.  Everything lands in
# gpurun_out/ev/; tools/collect_profiles.py r06 copies what is judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ev; rm -rf $O; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
rm -f $R/gpurun_out/parity_report.jsonl
timeout 1800 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" | tail -15 > $O/pytest_gpu.log
cp $R/gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
( time timeout 1200 python bench.py ) 2>$O/bench_default.err | tail -1 > $O/bench_default.json
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2>$O/bench_driver_window.err | tail -1 > $O/bench_driver_window.json
timeout 600 python bench.py --config config1 --no-extra 2>/dev/null | tail -1 > $O/bench_config1.json
timeout 600 python bench.py --config abc800 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $O/bench_abc800.json
timeout 600 python bench.py --config config2 --init-opacity --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $O/bench_config2_init_opacity.json
for c in config3 config4; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $O/bench_${c}.json
done
timeout 300 python bench.py --force-dp --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 > $O/bench_config2_force_dp.json
timeout 300 python tools/train_abc_fixture.py 2>/dev/null | tail -5 > $O/train_abc_fixture.txt
timeout 300 python tools/late_epoch_bench.py 2>/dev/null | grep -v "$F" > $O/late_epoch_bench.txt
cd /tmp && export TMPDIR=/tmp
for c in config1 config2 config2i config3 config4 abc800; do
  rm -rf /tmp/ev_$c /tmp/evsq_$c
  a="--config $c"; [ $c = config2i ] && a="--config config2 --init-opacity"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ev_$c -o r -- python $R/bench.py $a --steps 300 --warmup 20 --profile-only > /dev/null 2>$O/prof_$c.err
  python $R/tools/rocpd_summary.py /tmp/ev_$c/r_results.db $O/kernel_stats_$c.txt > /dev/null
  python $R/tools/timeline_gaps.py /tmp/ev_$c/r_results.db > $O/timeline_gaps_$c.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/evsq_$c -o q -- python $R/bench.py $a --steps 40 --warmup 5 --profile-only > /dev/null 2>$O/sq_$c.err
  python $R/tools/pmc_sq_summary.py $O/sq_counters_$c.txt /tmp/evsq_$c/q_results.db > /dev/null
done
cd $R
tail -4 $O/pytest_gpu.log; head -c 600 $O/bench_default.json; echo; tail -4 $O/bench_default.err; for c in config1 config2 config3 config4 abc800; do head -7 $O/kernel_stats_$c.txt | cut -c1-120; done
