#!/usr/bin/env python3
"""Where the host time in front of a window's first kernel goes (round 6): after a read-back the GPU idles until the first
step's first kernel is enqueued; with 20-step windows (the driver's bench command) that idle time is ~9 % of the window.
usage: python tools/window_host_profile.py [config2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", name != "config1")
tr.ensure_capacity()
V = bench.CONFIGS[name][1]
for rep in range(20):
    vs = [s % V for s in range(50)]
    tr.train_steps(vs, [whole] * 50)
tr.pop_loss()
K = 20
acc = {}
def tick(label, t0):
    t1 = time.perf_counter(); acc.setdefault(label, []).append(t1 - t0); return t1
REPS = 30
for rep in range(REPS):
    torch.cuda.synchronize()
    t = time.perf_counter(); t_start = t
    vs = [(rep * K + s) % V for s in range(K)]
    wm = [ratio(v) if s % 5 == 0 else whole for s, v in enumerate(vs)]
    t = tick("weight maps (4 draws)", t)
    tr._reserve_tags(K); t = tick("reserve tags", t)
    if not tr._journal:
        tr._snapshot()
    t = tick("snapshot", t)
    tr._journal.extend(("1", v, w, tr.epoch, tr.loss_scale) for v, w in zip(vs, wm)); t = tick("journal", t)
    a, va, wa = tr._steps_begin(vs, wm); t = tick("steps_begin (argument block)", t)
    from edgegaussians_amd._lib import call, ptr, stream
    import ctypes as C
    call("eg_train_steps", C.byref(a), K, va, wa, ptr(tr.viewmats), ptr(tr.Ks), ptr(tr.gt), stream()); t = tick("native enqueue of 20 steps", t)
    tr._steps_end(K); t = tick("steps_end", t)
    torch.cuda.synchronize(); t = tick("wait for the GPU", t)
    acc.setdefault("window", []).append(t - t_start)
    tr.pop_loss()
for k, v in acc.items():
    v = sorted(v)
    print(f"{k:34s} median {1e6 * v[len(v) // 2]:9.1f}   max {1e6 * v[-1]:10.1f} us per window")
