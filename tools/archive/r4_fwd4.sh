#!/bin/bash
# Round 4, forward: front-slices-first dispatch on large tile grids (EG_FLAG_FRONT_PREFIX) A/B.  Development build.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-d}; mkdir -p $O; cd $R
export EG_DEV_SWITCHES=1
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "$F" | tail -15 > $O/pytest_$TAG.log
b() { env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json; }
b c4 "X=1" "--config config4"
b c4_itemorder "EG_FRONT_LARGE=0" "--config config4"
b c4_nogate "EG_WAVE_GATE_MIN=1000000" "--config config4"
b c3 "X=1" "--config config3"
b c3_itemorder "EG_FRONT_LARGE=0" "--config config3"
b c3_nogate "EG_WAVE_GATE_MIN=1000000" "--config config3"
b c4i "X=1" "--config config4 --init-opacity"
b c3i "X=1" "--config config3 --init-opacity"
b c2s "X=1" "--config config2"
b c2s_f9 "EG_FRONT_SLICES=9" "--config config2"
b c2s_f2 "EG_FRONT_SLICES=2" "--config config2"
b c2i_f9 "EG_FRONT_SLICES=9" "--config config2 --init-opacity"
EG_FWD_PROF=1 timeout 300 python tools/fwd_prof.py config4 --spread 2>/dev/null | grep -v "$F" > $O/fwd_phases_c4_$TAG.txt
timeout 300 python tools/fwd_sched_sim.py config4 --spread 2>/dev/null | grep -v "$F" > $O/sched_sim_c4_$TAG.txt
cat $O/pytest_$TAG.log | tail -5
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', d['config'].get('tile_intersections_M'), {k:round(v,1) for k,v in d.get('stages_us',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/fwd_phases_c4_$TAG.txt $O/sched_sim_c4_$TAG.txt
