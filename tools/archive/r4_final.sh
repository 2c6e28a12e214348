#!/bin/bash
# Round 4, last checks: smoke(), parity seed sweep, roctx ranges under rocprofv3 --marker-trace, the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-z}; mkdir -p $O; cd $R
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 600 python __graft_entry__.py --smoke 2>&1 | grep -v "$F" | tail -3 > $O/smoke_$TAG.txt
timeout 1500 python tools/parity_seed_sweep.py 2>&1 | grep -v "$F" | tail -30 > $O/parity_seed_sweep_$TAG.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ev_roctx
timeout 300 rocprofv3 --kernel-trace --marker-trace --stats -d /tmp/ev_roctx -o r -- python $R/bench.py --config config2 --roctx --steps 100 --warmup 10 --profile-only > /dev/null 2>$O/prof_roctx.err
python - > $O/roctx_ranges_config2_$TAG.txt 2>&1 <<PY
import sqlite3, glob
db = glob.glob("/tmp/ev_roctx/**/*_results.db", recursive=True)
cur = sqlite3.connect(db[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("rocprofv3 --kernel-trace --marker-trace over bench.py --config config2 --roctx --steps 100")
for t in ("regions", "rocpd_region", "markers"):
    if t in tabs:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        print(t, "columns:", cols)
        nm = next((c for c in ("name", "region_name", "message") if c in cols), None)
        if nm and "start" in cols and "end" in cols:
            agg = {}
            for n, s0, e0 in cur.execute(f"select {nm}, start, end from {t}"):
                if str(n).startswith("eg:"):
                    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e0 - s0) / 1e3
            for n, a in sorted(agg.items()):
                print(f"{n:24s} ranges {a[0]:6d}  mean host span {a[1] / a[0]:8.2f} us")
            if agg:
                break
PY
cd $R
( time timeout 900 python bench.py ) 2>$O/bench_default_$TAG.err | tail -1 > $O/bench_default_$TAG.json
cat $O/smoke_$TAG.txt $O/parity_seed_sweep_$TAG.txt $O/roctx_ranges_config2_$TAG.txt; tail -4 $O/bench_default_$TAG.err
python - <<PY
import json
d=json.loads(open("$O/bench_default_$TAG.json").read())
print(d["ms_per_step"], d["value"], d.get("value_config1"), d.get("ms_per_step_config1"), d.get("ms_per_step_windows_config1"), d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
