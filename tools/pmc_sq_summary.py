#!/usr/bin/env python3
"""Per-kernel averages of arbitrary PMC counters from rocprofv3 rocpd databases.
usage: pmc_sq_summary.py out.txt db1 [db2 ...]"""
import collections
import sqlite3
import sys

out = open(sys.argv[1], "w")
for path in sys.argv[2:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k, c, v in rows:
        s = k.split("(")[0].replace("void ", "").replace("eg::", "").split("<")[0]
        a = agg[(s, c)]
        a[0] += 1
        a[1] += v
    for k in sorted(set(k for k, _ in agg)):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        line = k + " " + str({c: round(a[1] / a[0]) for (kk, c), a in sorted(agg.items()) if kk == k})
        print(line)
        out.write(line + "\n")
