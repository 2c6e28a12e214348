#!/usr/bin/env python3
"""round 6, development build -DEG_BF_PROF: where a workgroup of the fused backward kernel (csrc/backward_fused.hip) spends
its time.  Every workgroup's first wave leaves eight wall-clock stamps (100 MHz) in the g2d buffer; this script runs a few
steps and prints, per phase, the median / 90th percentile duration over the workgroups and the launch's own span.
usage: EG_EXTRA_HIPCC_FLAGS="-DEG_BF_PROF" python -m edgegaussians_amd.build --force; python tools/bf_prof.py config2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", name != "config1")
tr.ensure_capacity()
tr.g2d = torch.zeros(2 * tr.N, 8, device="cuda")  # (the stamps land BEHIND the [N, 8] records, which the run's last step -- two kernels -- overwrites)
tr._args_cache = {}
V = bench.CONFIGS[name][1]
views = [s % V for s in range(50)]
for rep in range(10):
    tr.train_steps(views, [whole] * 50)
tr.pop_loss()
torch.cuda.synchronize()
nb = (tr.N + 63) // 64
names = ["sized+walk+reduce (own wave)", "wait for the other waves", "loads + fwd recompute + VJP", "moments + Adam + stores",
         "next projection + tile tests + hist", "slot atomics", "key stores"]
rows = []
for rep in range(5):
    tr.g2d.zero_()
    tr.train_steps(views[:2], [whole] * 2)   # step 0: the fused kernel; step 1, the run's last, has no next view: two kernels
    torch.cuda.synchronize()
    st = tr.g2d[tr.N:].contiguous().view(torch.int64).view(-1)[: nb * 16].view(nb, 16)[:, :8].cpu().numpy().astype(np.float64) * 0.01  # us
    if (st[:, 0] == 0).any():
        continue
    rows.append(st)
st = rows[-1]
t0 = st[:, 0].min()
print(f"{name}: N {tr.N}, {nb} workgroups; launch span (first start -> last end) {st[:, 7].max() - t0:.2f} us; "
      f"last phase-1 end {st[:, 2].max() - t0:.2f} us; workgroup start times: median {np.median(st[:, 0] - t0):.2f}, max {(st[:, 0] - t0).max():.2f} us")
d = np.diff(st, axis=1)
for i, n_ in enumerate(names):
    print(f"  {n_:40s} median {np.median(d[:, i]):7.2f}  p90 {np.quantile(d[:, i], 0.9):7.2f}  max {d[:, i].max():7.2f} us")
print(f"  {'phase 2 in all':40s} median {np.median(st[:, 7] - st[:, 2]):7.2f}  p90 {np.quantile(st[:, 7] - st[:, 2], 0.9):7.2f}  max {(st[:, 7] - st[:, 2]).max():7.2f} us")
late = np.argsort(st[:, 7])[-5:]
print("  the five workgroups that end last: index, start, phase-1 end, end (us after the first start):")
for b in late:
    print(f"    {b:6d} {st[b, 0] - t0:8.2f} {st[b, 2] - t0:8.2f} {st[b, 7] - t0:8.2f}")
