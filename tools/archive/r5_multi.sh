#!/bin/bash
# round 5: sort phase profile (development build) + the native multi-scene call (test, bench legs)
#   gpurun --timeout 1500 -- 'bash tools/r5_multi.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5multi; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
if [ "${SORTPROF:-1}" = "1" ]; then
EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF" python -m edgegaussians_amd.build --force 2>&1 | tail -1
for c in config1 config2; do timeout 300 python tools/sort_prof.py $c --spread 2>&1 | grep -v "$F" > $O/sort_phases_$c.txt; cat $O/sort_phases_$c.txt; done
timeout 300 python tools/sort_prof.py config3 --spread 2>&1 | grep -v "$F" > $O/sort_phases_config3.txt; head -14 $O/sort_phases_config3.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
fi
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "multi_scene or side_by_side" 2>&1 | grep -v "$F" | tail -5
for c in ${CFGS:-config1 config2}; do for S in ${SS:-1 2 4 8}; do
  [ $c = config2 ] && [ $S = 8 ] && continue
  for drv in "threads 0" "native 1" "native 2" "native 0"; do set -- $drv
    timeout 300 python bench.py --config $c --scenes-per-gpu $S --scenes-driver $1 --scenes-threads $2 2>/dev/null | tail -1 > $O/scenes_${c}_S${S}_$1$2.json
    python - <<PY
import json
try:
    d = json.loads(open("$O/scenes_${c}_S${S}_$1$2.json").read())
    print("$c S=$S $1 threads=$2: %7.0f M Gv/s  %6.1f us per scene-step  host %6.1f us/step" % (d["value"] / 1e6, d["aggregate_us_per_scene_step"], 1e3 * d["host_enqueue_ms_per_step"]))
except Exception as e:
    print("$c S=$S $1 $2: ERR", e)
PY
    [ $S = 1 ] && break
  done
done; done 2>&1 | tee $O/scenes_summary.txt
