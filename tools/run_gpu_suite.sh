#!/bin/bash
# Full evidence run on the GPU box: parity tests, bench lines, rocprofv3 kernel stats, PMC traffic.
#   gpurun --timeout 2400 -- 'bash tools/run_gpu_suite.sh'
# Everything lands in gpurun_out/; copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $O/pytest_gpu.log
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
for c in config1 config3 config4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${c}.json
done
timeout 600 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_spread.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r01 -- python $R/bench.py --steps 200 --warmup 20 --profile-only > /dev/null 2>$O/prof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o f -- python $R/bench.py --steps 40 --warmup 5 --profile-only > /dev/null 2>$O/pmc_f.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o w -- python $R/bench.py --steps 40 --warmup 5 --profile-only > /dev/null 2>$O/pmc_w.err
cd $R
python tools/rocpd_summary.py /tmp/prof_c2/r01_results.db $O/kstats.txt > /dev/null
python tools/timeline_gaps.py /tmp/prof_c2/r01_results.db > $O/gaps.txt
python tools/pmc_summary.py /tmp/pmc_fetch/f_results.db /tmp/pmc_write/w_results.db $O/pmc_config2.json > /dev/null
# second bench pass so that roofline.traffic is filled from the PMC file of THIS build
cp $O/pmc_config2.json $R/profiles/r01_pmc_config2.json
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
tail -3 $O/pytest_gpu.log; head -c 600 $O/bench_c2.json; echo; head -20 $O/kstats.txt
