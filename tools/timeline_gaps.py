#!/usr/bin/env python3
"""Per-kernel duration AND the idle gap before each kernel, averaged over the steady-state steps of a
rocprofv3 --kernel-trace database: where the step's wall time goes (kernel time vs launch boundaries).
usage: timeline_gaps.py <results.db> [skip_first_n_dispatches]"""
import sqlite3
import sys
from collections import OrderedDict


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
    rows = rows[skip:]
    agg = OrderedDict()
    prev_end = None
    for name, s, e in rows:
        short = name.split("(")[0].replace("void ", "").replace("eg::", "")[:48]
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        if prev_end is not None:
            a[2] += (s - prev_end) / 1e3
        prev_end = e
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"{'kernel':50s} {'calls':>6s} {'avg_us':>8s} {'gap_before_us':>14s}")
    tk = tg = 0.0
    for k, a in agg.items():
        print(f"{k:50s} {a[0]:6d} {a[1] / a[0]:8.2f} {a[2] / a[0]:14.2f}")
        tk += a[1]
        tg += a[2]
    print(f"span {span:.1f} us: kernels {tk:.1f} us ({100 * tk / span:.1f} %), gaps {tg:.1f} us ({100 * tg / span:.1f} %)")


if __name__ == "__main__":
    main()
