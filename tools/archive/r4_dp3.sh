#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "native_data_parallel or data_parallel_leg" 2>&1 | grep -v "$F" | tail -3
for c in config2 config1; do timeout 400 python bench.py --config $c --force-dp --no-cpu-baseline --no-extra --no-traffic 2>/dev/null | tail -1 > $O/bench_${c}_dp_final.json; done
python - <<PY
import json
for c in ("config2","config1"):
    d=json.loads(open("$O/bench_%s_dp_final.json" % c).read())
    print(c, round(d['ms_per_step']*1e3,1),'us host_enqueue', round(d['host_enqueue_ms_per_step']*1e3,1), d.get('native_dp_host_us_per_step'), d['config'].get('data_parallel_leg'))
PY
