#!/usr/bin/env python3
"""Per-phase cost of the tile sort kernel's workgroups (development build with -DEG_SORT_PROF): shader-clock ticks per
phase of every tile's workgroup of the LAST launch, the wall-clock window of the launch and the slowest workgroups.
usage: python tools/sort_prof.py [config2] [--spread]   (EG_EXTRA_HIPCC_FLAGS=-DEG_SORT_PROF python -m edgegaussians_amd.build --force first)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from edgegaussians_amd import _lib  # noqa: E402

name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
spread = "--spread" in sys.argv
tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", spread)
tr.ensure_capacity()
V = bench.CONFIGS[name][1]
for rep in range(3):
    tr.train_steps([s % V for s in range(50)], [whole] * 50)
    tr.pop_loss()
lib = _lib.load()
lib.eg_debug_sort_profile.restype = C.c_int64
lib.eg_debug_sort_profile.argtypes = [C.c_void_p, C.c_int64]
tr.train_steps([s % V for s in range(50)], [whole] * 50)
tr.pop_loss()
torch.cuda.synchronize()
T = tr.T
buf = np.zeros((T, 12), np.uint64)
n = lib.eg_debug_sort_profile(buf.ctypes.data, T)
assert n > 0, n
rec = buf[:n].astype(np.float64)
ran = rec[:, 1] > 0
rec = rec[ran]
tiles = np.nonzero(ran)[0]
names = ["loads + reductions", "range barrier + prefix", "histogram", "scan", "scatter", "rank", "store drain"]
w0, w1 = rec[:, 0].min(), rec[:, 1].max()
print(f"{name} spread={spread} M={tr.last_m()} tiles={len(rec)} launch window {(w1 - w0) / 100.0:.2f} us (100 MHz wall clock, first start -> last end)")
pop = rec[:, 2]
life = rec[:, 3:10].sum(axis=1)
print(f"  populations: mean {pop.mean():.0f} median {np.median(pop):.0f} p95 {np.percentile(pop, 95):.0f} max {pop.max():.0f}; "
      f"empty {int((pop <= 0).sum())}, <= 64: {int((pop <= 64).sum())}, <= 512: {int((pop <= 512).sum())}, > 1536: {int((pop > 1536).sum())}")
for i, nm in enumerate(names):
    c = rec[:, 3 + i]
    print(f"  {nm:24s} mean {c.mean():8.0f}  median {np.median(c):8.0f}  p95 {np.percentile(c, 95):8.0f}  max {c.max():8.0f} ticks")
print(f"  workgroup lifetime: mean {life.mean():.0f} median {np.median(life):.0f} p95 {np.percentile(life, 95):.0f} max {life.max():.0f} ticks"
      f" ({life.max() / 2400:.2f} us at 2.4 GHz); wall clock per workgroup mean {(rec[:, 1] - rec[:, 0]).mean() / 100:.2f} us max {(rec[:, 1] - rec[:, 0]).max() / 100:.2f} us")
st = (rec[:, 0] - w0) / 100.0
en = (rec[:, 1] - w0) / 100.0
print("  starts (us after the first): " + " ".join(f"p{q}={np.percentile(st, q):.2f}" for q in (10, 50, 75, 90, 99, 100)))
print("  ends   (us after the first start): " + " ".join(f"p{q}={np.percentile(en, q):.2f}" for q in (10, 50, 75, 90, 99, 100)))
order = np.argsort(-en)[:12]
print("  last workgroups to end: tile  n  start_us  end_us  | ticks per phase")
for i in order:
    print(f"    {tiles[i]:5d} {int(pop[i]):5d} {st[i]:7.2f} {en[i]:7.2f} | " + " ".join(f"{int(x):6d}" for x in rec[i, 3:10]))
for lo, hi in ((0, 0), (1, 64), (65, 512), (513, 1536), (1537, 4096)):
    m = (pop >= lo) & (pop <= hi)
    if m.any():
        print(f"  n in [{lo},{hi}]: {int(m.sum())} tiles, lifetime mean {life[m].mean():.0f} ticks, wall {(en[m] - st[m]).mean():.2f} us | "
              + " ".join(f"{rec[m, 3 + i].mean():6.0f}" for i in range(7)))

xcc, wg = rec[:, 10].astype(int), rec[:, 11].astype(int)
print(f"  XCC_ID of workgroup b: == b % 8 for {int((xcc == wg % 8).sum())} of {len(wg)} workgroups; per-XCC counts {np.bincount(xcc, minlength=8).tolist()}")

