#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "regularis or data_parallel or dist or knn or reduce_words" 2>&1 | grep -v "$F" | tail -15
timeout 300 python tools/late_epoch_bench.py 2>/dev/null | grep -v "$F"
