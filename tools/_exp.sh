cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 > gpurun_out/smoke.log
timeout 600 python bench.py --config config4 --no-cpu-baseline 2>gpurun_out/c4.err | tail -1 > gpurun_out/bench_config4.json
