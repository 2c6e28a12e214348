#!/bin/bash
# Iteration run on the GPU box: GPU tests (quick subset unless FULL=1), then bench lines of config1/2/2-trained-like.
#   gpurun --timeout 1500 -- 'TAG=x bash tools/gpu_iter.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-it}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
if [ "${TESTS:-1}" = "1" ]; then
  rm -f $O/parity_report.jsonl
  DESEL=""; [ "${FULL:-0}" = "1" ] || DESEL="--deselect tests/test_gpu_parity.py::test_end_to_end_training_recovers_ground_truth_edges"
  timeout 1400 python -m pytest tests -m gpu -q --tb=short ${XFLAG:--x} $DESEL 2>&1 | grep -v "$F" | tail -${TAIL:-40} > $O/pytest_$TAG.log
fi
for c in ${CONFIGS:-config1 config2}; do
  timeout 300 python bench.py --config $c --init-opacity --no-cpu-baseline --no-traffic --no-extra 2>$O/bench_${c}_$TAG.err | tail -1 > $O/bench_${c}_$TAG.json
done
[ "${SPREAD:-1}" = "1" ] && timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 > $O/bench_config2s_$TAG.json
for c in ${BATCHED:-config1 config2}; do for v in 2 4 8; do timeout 300 python bench.py --config $c --views-per-step $v --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-extra 2>$O/bench_${c}_b${v}_$TAG.err | tail -1 > $O/bench_${c}_b${v}_$TAG.json; done; done
tail -${TAIL:-40} $O/pytest_$TAG.log 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', round(d['value']/1e6),'MGv/s', d['config']['tile_intersections_M'], {k:round(v,1) for k,v in d.get('stages_us',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
if [ "${OPER:-0}" = "1" ]; then for c in config1 config2; do timeout 300 python bench.py --config $c --path operator 2>$O/bench_${c}_oper_$TAG.err | tail -1 > $O/bench_${c}_oper_$TAG.json; python -c "
import json; d=json.loads(open('$O/bench_${c}_oper_$TAG.json').read()); print('$c operator', round(d['ms_per_step']*1e3,1),'us')"; done; fi
if [ "${ABC:-0}" = "1" ]; then timeout 600 python tools/train_abc_fixture.py > $O/abc_$TAG.txt 2>&1; tail -5 $O/abc_$TAG.txt | cut -c1-400; fi
