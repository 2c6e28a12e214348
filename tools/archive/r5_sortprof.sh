#!/bin/bash
# round 5: per-phase profile of the sort kernel's workgroups (development build, -DEG_SORT_PROF)
#   gpurun --timeout 900 -- 'bash tools/r5_sortprof.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5sortprof; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
EG_EXTRA_HIPCC_FLAGS="-DEG_SORT_PROF $XFLAGS" python -m edgegaussians_amd.build --force 2>&1 | tail -1
for c in ${CFGS:-config1 config2}; do timeout 300 python tools/sort_prof.py $c --spread 2>&1 | grep -v "$F" > $O/sort_phases_$c.txt; cat $O/sort_phases_$c.txt; done
python -m edgegaussians_amd.build --force 2>&1 | tail -1
