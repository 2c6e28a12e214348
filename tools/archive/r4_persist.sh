#!/bin/bash
# Round 4: persistent launch of the forward, A/B (development build).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-p}; mkdir -p $O; cd $R
export EG_DEV_SWITCHES=1
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "chained_forward or config2_real_poses or fullsize or speculation or batched or dispatch_order or operator_fast or smoke or native_run" 2>&1 | grep -v "$F" | tail -8 > $O/pytest_$TAG.log
b() { env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json; }
b c2s "X=1" "--config config2"
b c2s_nop "EG_WAVE_PERSIST=0" "--config config2"
b c2i "X=1" "--config config2 --init-opacity"
b c2i_nop "EG_WAVE_PERSIST=0" "--config config2 --init-opacity"
b c1 "X=1" "--config config1"
b c1_nop "EG_WAVE_PERSIST=0" "--config config1"
b c4 "X=1" "--config config4"
b c4_nop "EG_WAVE_PERSIST=0" "--config config4"
b c3 "X=1" "--config config3"
b c3_nop "EG_WAVE_PERSIST=0" "--config config3"
tail -6 $O/pytest_$TAG.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', {k:round(v,1) for k,v in d.get('stages_us',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
