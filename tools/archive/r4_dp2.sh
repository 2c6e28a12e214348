#!/bin/bash
# Round 4: full GPU test-suite + launch gaps of the data-parallel leg on one GPU, native vs Python driver.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-i}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
rm -f $R/gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" | tail -30 > $O/pytest_$TAG.log
cp $R/gpurun_out/parity_report.jsonl $O/parity_report_$TAG.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for m in native python fused; do
  rm -rf /tmp/dp_$m; e="X=1"; a="--force-dp"; [ $m = python ] && e="EG_NO_NATIVE_DP=1"; [ $m = fused ] && a=""
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dp_$m -o r -- python $R/bench.py --config config2 $a --steps 300 --warmup 20 --profile-only > /dev/null 2>$O/prof_dp_$m.err
  python $R/tools/timeline_gaps.py /tmp/dp_$m/r_results.db > $O/timeline_gaps_dp_${m}_$TAG.txt
done
cd $R
tail -12 $O/pytest_$TAG.log
for m in native python fused; do echo "--- $m"; cat $O/timeline_gaps_dp_${m}_$TAG.txt; done
