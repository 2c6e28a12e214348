#!/usr/bin/env python3
"""Per-view launch durations of the step's kernels from a rocprofv3 --kernel-trace database of `bench.py --profile-only`
(the views cycle 0 .. V-1 in launch order): which views are the expensive ones.  usage: kernel_per_view.py <results.db> [V=50]"""
import sqlite3, sys
import numpy as np
cur = sqlite3.connect(sys.argv[1]).cursor()
V = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
for key in ("composite_wave_fwd_kernel", "footprint_bwd_kernel", "tile_sort_kernel", "project_bwd_emit_kernel"):
    d = np.array([(e - s) / 1e3 for n, s, e in rows if key in n])
    d = d[len(d) % V:] if len(d) >= 2 * V else d
    n = len(d) // V * V
    if n == 0:
        continue
    per = d[-n:].reshape(-1, V).mean(axis=0)   # phase within the cycle (the view index up to a constant rotation)
    order = np.argsort(-per)
    print(f"{key}: mean {per.mean():.1f} us, median {np.median(per):.1f}, min {per.min():.1f}, max {per.max():.1f}; "
          f"the 5 most expensive cycle positions: " + ", ".join(f"{i}:{per[i]:.1f}" for i in order[:5]))
    print("   all:", " ".join(f"{x:.0f}" for x in per))
