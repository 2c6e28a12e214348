cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/ab_pytest.log
for c in config1 config2 config3 config4; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>gpurun_out/ab_err_$c.log | tail -1 > gpurun_out/ab_$c.json; done
timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_config2s.json
