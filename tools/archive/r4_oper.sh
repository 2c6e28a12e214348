#!/bin/bash
# Round 4: the drop-in operator over the two native calls (operator tests, host-time profile, operator bench lines) + the
# native data-parallel leg's host profile.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-g}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "operator or boundary or rasterization or reference_protocol or bench_line or kernel_trace or smoke or drop_in" 2>&1 | grep -v "$F" | tail -30 > $O/pytest_$TAG.log
timeout 300 python tools/operator_profile.py config2 2>/dev/null | grep -v "$F" > $O/operator_profile_config2_$TAG.txt
( echo "--- with edgegaussians_amd.optim.Adam ---"; timeout 300 python tools/operator_profile.py config2 native 2>/dev/null | grep -v "$F" ) >> $O/operator_profile_config2_$TAG.txt
for a in torch native; do timeout 300 python bench.py --path operator --operator-adam $a --no-cpu-baseline --no-traffic --no-extra 2>$O/bench_oper_${a}_$TAG.err | tail -1 > $O/bench_oper_${a}_$TAG.json; done
timeout 300 python bench.py --config config1 --path operator --operator-adam native --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 > $O/bench_oper_c1_native_$TAG.json
timeout 400 python bench.py --config config2 --force-dp --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_dp_$TAG.err | tail -1 > $O/bench_dp_$TAG.json
tail -30 $O/pytest_$TAG.log
cat $O/operator_profile_config2_$TAG.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', d.get('native_dp_host_us_per_step'), d.get('host_enqueue_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
