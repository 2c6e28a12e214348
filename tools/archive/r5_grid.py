#!/usr/bin/env python3
"""round 5: how much of a step is the forward's surplus workgroups?  The forward's grid covers max_items; sized tighter (capacity slack 1.3 ->
1.1 -> 1.02) fewer workgroups find no record.  usage: python tools/r5_grid.py [config2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
V = bench.CONFIGS[name][1]
for slack in (1.3, 1.1, 1.02, 1.3, 1.02):
    tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", name != "config1")
    tr.ensure_capacity(slack=slack)
    views = [s % V for s in range(50)]
    for rep in range(20):
        tr.train_steps(views, [whole] * 50)
    tr.pop_loss()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(40):
        tr.train_steps(views, [whole] * 50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr.pop_loss()
    print(f"{name} slack {slack}: max_items {tr.max_items} (items {int(tr.total.cpu()[2])}), {1e6 * dt / 2000:.2f} us/step, overflow events {tr.overflow_events}", flush=True)
