#!/bin/bash
# round 5: where the head of a sort workgroup goes (finer ticks) + kernel arguments in device memory (HIP_FORCE_DEV_KERNARG)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5karg; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF" python -m edgegaussians_amd.build --force 2>&1 | tail -1
for k in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$k"; HIP_FORCE_DEV_KERNARG=$k timeout 300 python tools/sort_prof.py config2 --spread 2>&1 | grep -v "$F" | grep -v "^    " > $O/sort_phases_config2_karg$k.txt; grep "window\|loads + red\|lifetime:\|head,\|the same" $O/sort_phases_config2_karg$k.txt; done
python -m edgegaussians_amd.build --force 2>&1 | tail -1
for k in 0 1; do for c in config1 config2; do
  HIP_FORCE_DEV_KERNARG=$k timeout 300 python bench.py --config $c --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 > $O/bench_${c}_karg$k.json
  python -c "
import json; d=json.loads(open('$O/bench_${c}_karg$k.json').read()); print('HIP_FORCE_DEV_KERNARG=$k $c', round(d['ms_per_step']*1e3,2),'us/step', {k:round(v,1) for k,v in d.get('kernel_trace_avg_us',{}).items()}, 'host', round(d['host_enqueue_ms_per_step']*1e3,1))"
done; done
