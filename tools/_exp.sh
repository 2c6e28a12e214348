cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do EG_SEGMENTED=1 timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/exp_s$i.json; done
for i in 1 2; do EG_SEGMENTED=0 timeout 300 python bench.py --config config2 --spread-opacity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/exp_c$i.json; done
