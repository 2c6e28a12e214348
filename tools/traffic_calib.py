#!/usr/bin/env python3
"""Counted / known bytes of rocprofv3's FETCH_SIZE and WRITE_SIZE (and the raw request counters behind them) for the
access patterns of the compositing forward: runs tools/microbench/traffic_calib under rocprofv3 --pmc, one pass per
counter set.   usage (GPU box): python tools/traffic_calib.py [out.txt]"""
import os, shutil, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "microbench", "traffic_calib")
rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
KNOWN = 64 << 20
sets = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_BUBBLE_sum"],
        ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]]
res = {}
for cs in sets:
    tmp = tempfile.mkdtemp(prefix="cal_", dir="/tmp")
    r = subprocess.run([rp, "--kernel-trace", "--pmc", *cs, "-d", tmp, "-o", "c", "--", exe], cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
    db = None
    for dp, _, fs in os.walk(tmp):
        for f in fs:
            if f.endswith("_results.db"):
                db = os.path.join(dp, f)
    if db is None:
        print("pass failed:", cs, r.stderr[-300:]); continue
    for k, c, v in sqlite3.connect(db).cursor().execute("select kernel_name, counter_name, value from counters_collection"):
        k = k.split("(")[0]
        a = res.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    shutil.rmtree(tmp, ignore_errors=True)
lines = [f"known bytes per launch: {KNOWN} (64 MiB; cal_store12: {KNOWN // 12 * 12})"]
for k in sorted(res):
    d = {c: v[1] / max(v[0], 1) for c, v in res[k].items()}
    f, w = d.get("FETCH_SIZE", 0) * 1024, d.get("WRITE_SIZE", 0) * 1024
    lines.append(f"{k:18s} FETCH_SIZE {f / KNOWN:6.3f} x known   WRITE_SIZE {w / KNOWN:6.3f} x known   " +
                 "  ".join(f"{c}={d[c]:.0f}" for c in sorted(d) if c.startswith("TCC")))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
