#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "binning or fullsize or big_footprints or large_tile or giant or huge or dispatch_order" 2>&1 | grep -v "$F" | tail -8
timeout 400 python bench.py --config config4 --no-cpu-baseline --no-extra --no-traffic 2>/dev/null | tail -1 > $O/bench_c4_sort.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks4
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks4 -o r -- python $R/bench.py --config config4 --steps 200 --warmup 20 --profile-only > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/ks4/r_results.db $O/kernel_stats_config4_sort.txt | head -8
python -c "
import json; d=json.loads(open('$O/bench_c4_sort.json').read()); print(round(d['ms_per_step']*1e3,1), {k:round(v,1) for k,v in d['stages_us'].items()})"
