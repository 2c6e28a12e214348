#!/usr/bin/env python3
"""Summarises the roctx ranges of a rocprofv3 --kernel-trace --marker-trace database (bench.py --roctx: eg_roctx_enable
wraps the stages of eg_train_step): count and mean host span per range name.   usage: roctx_summary.py <results.db> [out.txt]"""
import json, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = {}
for name, s0, e0, ext in cur.execute("select name, start, end, extdata from regions where category like 'MARKER%'"):
    try:
        msg = json.loads(ext).get("message", name)
    except Exception:
        msg = name
    a = agg.setdefault(msg, [0, 0.0])
    a[0] += 1
    a[1] += (e0 - s0) / 1e3
kern = {}
for name, s0, e0 in cur.execute("select name, start, end from kernels"):
    k = name.split("(")[0].replace("void ", "").replace("eg::", "").split("<")[0]
    a = kern.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e0 - s0) / 1e3
lines = ["roctx ranges (host spans of the enqueue of each stage of eg_train_step; rocprofv3 --kernel-trace --marker-trace over",
         "`bench.py --config config2 --roctx --profile-only`), next to the kernel trace of the same run:"]
for n, a in sorted(agg.items()):
    lines.append(f"  range  {n:24s} count {a[0]:6d}   mean host span {a[1] / a[0]:8.2f} us")
for n, a in sorted(kern.items(), key=lambda kv: -kv[1][1])[:6]:
    lines.append(f"  kernel {n:40s} count {a[0]:6d}   mean {a[1] / a[0]:8.2f} us")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
