#!/bin/bash
# round 6: (1) the failing bit-exactness case with its traceback, three times; (2) phase profile of the fused backward kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6prof; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
{
python -m edgegaussians_amd.build 2>&1 | grep -v "^built" | tail -2
for w in 4 8; do
  EG_EXTRA_HIPCC_FLAGS="-DEG_BF_PROF -DEG_BF_WAVES_PER_EU=$w" python -m edgegaussians_amd.build --force 2>&1 | grep -v "^built" | tail -2
  echo "=== waves per EU $w"
  for c in config1 config2; do timeout 300 python tools/bf_prof.py $c 2>&1 | grep -v "$F"; done
done
} 2>&1 | tee $O/summary.txt
python -m edgegaussians_amd.build --force 2>&1 | tail -1
