cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/fp_pytest.log
for c in 1 2 3; do
  timeout 300 python bench.py --config config$c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fp_c${c}_m2.json
done
timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fp_c2s_m2.json
