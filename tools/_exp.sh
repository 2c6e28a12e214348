cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_c2 -o r01 -- python $R/bench.py --steps 200 --warmup 20 --profile-only > /dev/null 2>$R/gpurun_out/prof.err
cd $R; python tools/timeline_gaps.py /tmp/prof_c2/r01_results.db > gpurun_out/gaps.txt; cat gpurun_out/gaps.txt
