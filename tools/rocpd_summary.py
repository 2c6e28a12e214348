#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace [+ PMC]) into a small text/JSON file
for profiles/.  Usage: rocpd_summary.py <results.db> [out.txt]"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.split("(")[0]
        short = short.replace("void ", "").replace("eg::", "")
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k[:70]:70s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/total:6.2f}")
    # PMC counters, if any
    try:
        pm = cur.execute("select * from pmc_events limit 1").fetchall()
        if pm:
            pcols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
            lines.append("")
            lines.append("PMC columns: " + ",".join(pcols))
    except sqlite3.Error:
        pass
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")




def tail_dispatches(path, n=200):
    """Debug helper: the last n kernel dispatches in order (name, duration us, gap to previous us)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()[-n:]
    prev = None
    for name, s, e in rows:
        short = name.split("(")[0].replace("void ", "").replace("eg::", "")[:50]
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{short:52s} {(e - s) / 1e3:9.2f} {gap:9.2f}")
        prev = e


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--tail":
        tail_dispatches(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 200)
    else:
        main()
