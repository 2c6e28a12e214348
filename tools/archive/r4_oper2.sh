#!/bin/bash
# Round 4: operator with deferred read-back; multi-scene mode; DP leg kernel stats native vs python.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-h}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "operator or boundary or reference_protocol or scenes_side or roctx or drop_in or smoke" 2>&1 | grep -v "$F" | tail -30 > $O/pytest_$TAG.log
for d in 1 0; do
( echo "=== EG_OPERATOR_DEFER=$d, torch.optim.Adam ==="; EG_OPERATOR_DEFER=$d timeout 300 python tools/operator_profile.py config2 2>/dev/null | grep -v "$F"
  echo "=== EG_OPERATOR_DEFER=$d, edgegaussians_amd.optim.Adam ==="; EG_OPERATOR_DEFER=$d timeout 300 python tools/operator_profile.py config2 native 2>/dev/null | grep -v "$F" ) >> $O/operator_profile_config2_$TAG.txt
done
for a in torch native; do timeout 300 python bench.py --path operator --operator-adam $a --no-cpu-baseline --no-traffic --no-extra 2>$O/bench_oper_${a}_$TAG.err | tail -1 > $O/bench_oper_${a}_$TAG.json; done
for S in 1 2 4 8; do timeout 300 python bench.py --config config1 --scenes-per-gpu $S 2>$O/bench_scenes${S}_$TAG.err | tail -1 > $O/bench_scenes${S}_c1_$TAG.json; done
for S in 1 2 4; do timeout 300 python bench.py --config config2 --scenes-per-gpu $S 2>/dev/null | tail -1 > $O/bench_scenes${S}_c2_$TAG.json; done
cd /tmp && export TMPDIR=/tmp
for m in native python; do
  rm -rf /tmp/dp_$m; e="X=1"; [ $m = python ] && e="EG_NO_NATIVE_DP=1"
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dp_$m -o r -- python $R/bench.py --config config2 --force-dp --steps 300 --warmup 20 --profile-only > /dev/null 2>$O/prof_dp_$m.err
  python $R/tools/rocpd_summary.py /tmp/dp_$m/r_results.db $O/kernel_stats_dp_${m}_$TAG.txt > /dev/null
done
cd $R
tail -30 $O/pytest_$TAG.log; cat $O/operator_profile_config2_$TAG.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1) if 'ms_per_step' in d else '', 'us', round(d['value']/1e6),'MGv/s', d.get('aggregate_us_per_scene_step'), d.get('host_enqueue_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
head -12 $O/kernel_stats_dp_native_$TAG.txt; head -12 $O/kernel_stats_dp_python_$TAG.txt
