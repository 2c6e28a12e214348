#!/usr/bin/env python3
"""Per-phase cost of the wave-autonomous forward (EG_FWD_PROF=1 builds the timed instantiation): shader-clock ticks per
phase, averaged over the waves of ~100 steps.  usage: EG_FWD_PROF=1 python tools/fwd_prof.py [config2] [--spread]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EG_FWD_PROF", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from edgegaussians_amd import _lib  # noqa: E402

name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
spread = "--spread" in sys.argv
tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", spread)
tr.ensure_capacity()
V = bench.CONFIGS[name][1]
for rep in range(3):
    views = [s % V for s in range(50)]
    tr.train_steps(views, [whole] * 50)
    tr.pop_loss()
lib = _lib.load()
lib.eg_debug_fwd_profile.restype = C.c_int64
lib.eg_debug_fwd_profile.argtypes = [C.c_void_p, C.c_int64]
import numpy as np  # noqa: E402
tr.train_steps([s % V for s in range(50)], [whole] * 50)
tr.pop_loss()
torch.cuda.synchronize()
cap = tr.max_items * 4
buf = np.zeros((cap, 8), np.uint64)
n = lib.eg_debug_fwd_profile(buf.ctypes.data, cap)
assert n > 0, n
rec = buf[:n][buf[:n, 7] == 1].astype(np.float64)
names = ["head (tables)", "staging", "walk", "publish", "ticket / look-back", "combine / exact stop", "epilogue"]
tot = rec[:, :7].sum()
life = rec[:, :7].sum(axis=1)
print(f"{name} spread={spread} M={tr.last_m()} waves={len(rec)} mode={'chained' if tr.rewalk_hint != 0 else 'speculative'}")
for i, nm in enumerate(names):
    c = rec[:, i]
    print(f"  {nm:24s} mean {c.mean():8.0f}  median {np.median(c):8.0f}  p95 {np.percentile(c, 95):8.0f}  max {c.max():8.0f} ticks  {100.0 * c.sum() / tot:5.1f} %")
print(f"  wave lifetime: mean {life.mean():.0f} median {np.median(life):.0f} p95 {np.percentile(life, 95):.0f} max {life.max():.0f} ticks; "
      f"sum {tot / 1e6:.1f} M ticks = {tot / 8192 / 2400:.2f} us x 8192 wave slots at 2.4 GHz")
busy = rec[rec[:, 2] > 0]
print(f"  waves with a non-empty list: {len(busy)}; their walk mean {busy[:, 2].mean():.0f}, staging mean {busy[:, 1].mean():.0f}")
