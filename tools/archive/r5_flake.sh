#!/bin/bash
# round 5: repeat the forward / binning test subset to catch an intermittent failure
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5flake; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_floats.py -q -k "chained or stop or dispatch_order or speculat or oracle_floats or binning" --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > $O/run_$i.log
  tail -1 $O/run_$i.log
  grep -q "failed" $O/run_$i.log && { grep -n "^FAILED\|^E " $O/run_$i.log | head -20; }
done
