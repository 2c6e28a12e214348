#!/bin/bash
# round 5, second evidence pass: the GPU suite on the final tree, the scenes-per-GPU legs, the sort kernel's phase record
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ev2; rm -rf $O; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
rm -f $R/gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "$F" | tail -15 > $O/pytest_gpu.log
cp $R/gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
tail -4 $O/pytest_gpu.log | cut -c1-200
for S in 1 2 4 8; do timeout 300 python bench.py --config config1 --scenes-per-gpu $S 2>/dev/null | tail -1 > $O/bench_config1_scenes$S.json; done
timeout 300 python bench.py --config config1 --scenes-per-gpu 8 --scenes-driver threads 2>/dev/null | tail -1 > $O/bench_config1_scenes8_threads.json
timeout 300 python bench.py --config config1 --scenes-per-gpu 8 --scenes-threads 1 2>/dev/null | tail -1 > $O/bench_config1_scenes8_native1.json
timeout 300 python bench.py --config config2 --scenes-per-gpu 4 2>/dev/null | tail -1 > $O/bench_config2_scenes4.json
timeout 300 python bench.py --config config2 --scenes-per-gpu 4 --scenes-driver threads 2>/dev/null | tail -1 > $O/bench_config2_scenes4_threads.json
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_config*_scenes*.json")):
    try:
        x=json.loads(open(f).read()); print(os.path.basename(f), round(x['value']/1e6),'M', round(x['aggregate_us_per_scene_step'],1),'us/scene-step host', round(1e3*x['host_enqueue_ms_per_step'],1), '|', x.get('driver'))
    except Exception as e: print(f,'ERR',e)
PY
EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
for c in config2 config1; do timeout 300 python tools/sort_prof.py $c --spread 2>&1 | grep -v "$F" > $O/sort_phases_$c.txt; grep -v "^    " $O/sort_phases_$c.txt | head -12; done
python -m edgegaussians_amd.build --force 2>&1 | tail -1
