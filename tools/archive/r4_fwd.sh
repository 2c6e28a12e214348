#!/bin/bash
# Round 4, forward hand-over rebuild: GPU tests, then bench lines with / without waiting for the anchor.
#   (build with EG_DEV_SWITCHES=1 for the A/B legs)   gpurun --timeout 1800 -- 'TAG=a bash tools/r4_fwd.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-a}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
if [ "${TESTS:-1}" = "1" ]; then
  rm -f $R/gpurun_out/parity_report.jsonl
  timeout 1200 python -m pytest tests -m gpu -q --tb=short ${XFLAG:--x} ${PYTEST_ARGS:-} 2>&1 | grep -v "$F" | tail -${TAIL:-30} > $O/pytest_$TAG.log
  cp $R/gpurun_out/parity_report.jsonl $O/parity_report_$TAG.jsonl 2>/dev/null
fi
b() { # name, env, args
  env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json
}
b c2s "X=1" "--config config2 --no-traffic"
b c2s_gate "EG_WAVE_GATE_MIN=0" "--config config2 --no-traffic"
b c2i "X=1" "--config config2 --init-opacity --no-traffic"
b c1 "X=1" "--config config1 --no-traffic"
b c4 "X=1" "--config config4 --no-traffic"
b c4_nogate "EG_WAVE_GATE_MIN=1000000" "--config config4 --no-traffic"
b c3 "X=1" "--config config3 --no-traffic"
b c3_nogate "EG_WAVE_GATE_MIN=1000000" "--config config3 --no-traffic"
if [ "${TRAFFIC:-1}" = "1" ]; then b c2s_traffic "X=1" "--config config2"; fi
EG_FWD_PROF=1 timeout 300 python tools/fwd_prof.py config2 --spread 2>/dev/null | grep -v "$F" > $O/fwd_phases_$TAG.txt
tail -${TAIL:-30} $O/pytest_$TAG.log 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read()); r=d.get('roofline',{})
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', round(d['value']/1e6),'MGv/s', d['config'].get('tile_intersections_M'), {k:round(v,1) for k,v in d.get('stages_us',{}).items()}, 'traffic', r.get('traffic'), 'alg', r.get('achieved'), r.get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/fwd_phases_$TAG.txt
