#!/bin/bash
# Round 4: full GPU tests + the data-parallel leg on one GPU, native against the Python driver.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-f}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
rm -f $R/gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short ${XFLAG:-} 2>&1 | grep -v "$F" | tail -40 > $O/pytest_$TAG.log
cp $R/gpurun_out/parity_report.jsonl $O/parity_report_$TAG.jsonl 2>/dev/null
b() { env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json; }
b c2s "X=1" "--config config2"
b c2s_dp_native "X=1" "--config config2 --force-dp"
b c2s_dp_python "EG_NO_NATIVE_DP=1" "--config config2 --force-dp"
b c1_dp_native "X=1" "--config config1 --force-dp"
tail -30 $O/pytest_$TAG.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us host_enqueue', round(d['host_enqueue_ms_per_step']*1e3,1), d['config'].get('data_parallel_leg'), d.get('allreduce_exposed_us_per_step_by_rank'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
