#!/bin/bash
# Round 4, forward: dispatch classes / anchors A/B, schedule simulation, counter calibration.  Development build.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-b}; mkdir -p $O; cd $R
export EG_DEV_SWITCHES=1
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "dispatch_order or chained_forward or config2_real_poses or fused_step_vs_c_oracle_full_size or speculation" 2>&1 | grep -v "$F" | tail -15 > $O/pytest_$TAG.log
b() { env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json; }
b c2s "X=1" "--config config2"
b c2s_nosingles "EG_SINGLES_LAST=0" "--config config2"
b c2s_noanchor "EG_WAVE_ANCHOR=0" "--config config2"
b c2s_r3like "EG_WAVE_ANCHOR=0 EG_SINGLES_LAST=0" "--config config2"
b c2i "X=1" "--config config2 --init-opacity"
b c2i_nosingles "EG_SINGLES_LAST=0" "--config config2 --init-opacity"
b c1 "X=1" "--config config1"
b c1_nosingles "EG_SINGLES_LAST=0" "--config config1"
b c4 "X=1" "--config config4"
b c4_noanchor "EG_WAVE_ANCHOR=0" "--config config4"
b c3_noanchor "EG_WAVE_ANCHOR=0" "--config config3"
timeout 300 python tools/fwd_sched_sim.py config2 --spread 2>/dev/null | grep -v "$F" > $O/sched_sim_$TAG.txt
timeout 300 python tools/fwd_sched_sim.py config4 --spread 2>/dev/null | grep -v "$F" >> $O/sched_sim_$TAG.txt
EG_FWD_PROF=1 timeout 300 python tools/fwd_prof.py config4 --spread 2>/dev/null | grep -v "$F" > $O/fwd_phases_c4_$TAG.txt
timeout 600 python tools/traffic_calib.py $O/traffic_calib_$TAG.txt > /dev/null 2>&1
cat $O/pytest_$TAG.log | tail -5
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read()); r=d.get('roofline',{})
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', d['config'].get('tile_intersections_M'), {k:round(v,1) for k,v in d.get('stages_us',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/sched_sim_$TAG.txt $O/fwd_phases_c4_$TAG.txt $O/traffic_calib_$TAG.txt
