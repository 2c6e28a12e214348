cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/exp_c2s.json
