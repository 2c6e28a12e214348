cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "spatial or training_loop or densify or io" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/exp_pytest.log
timeout 300 python bench.py --config config2 --no-cpu-baseline 2>gpurun_out/exp_err.log | tail -1 > gpurun_out/exp_config2.json
