#!/usr/bin/env python3
"""What the dispatch ORDER of the forward's workgroups is worth: takes the per-wave lifetimes of one real launch
(EG_FWD_PROF=1 phase records, indexed by record = dispatch index) and replays them through a list scheduler with 2048
workgroup slots (8 per CU x 256 CUs) in several orders.  Lifetimes are taken as measured (the timed build waits at every
tick, and a wave's waiting depends on the order, so this is a first-order estimate).
usage (GPU box): EG_FWD_PROF=1 python tools/fwd_sched_sim.py [config2] [--spread]"""
import ctypes as C, heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EG_FWD_PROF", "1")
import numpy as np, torch  # noqa: E402
import bench  # noqa: E402
from edgegaussians_amd import _lib  # noqa: E402
name = next((a for a in sys.argv[1:] if a.startswith("config")), "config2")
spread = "--spread" in sys.argv
tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", spread)
tr.ensure_capacity()
V = bench.CONFIGS[name][1]
for rep in range(4):
    tr.train_steps([s % V for s in range(50)], [whole] * 50); tr.pop_loss()
torch.cuda.synchronize()
lib = _lib.load()
lib.eg_debug_fwd_profile.restype = C.c_int64
lib.eg_debug_fwd_profile.argtypes = [C.c_void_p, C.c_int64]
cap = tr.max_items * 4
buf = np.zeros((cap, 8), np.uint64)
n = lib.eg_debug_fwd_profile(buf.ctypes.data, cap)
rec = buf[:n].astype(np.float64).reshape(-1, 4, 8)
tab = tr.item_rec.cpu().numpy()[:len(rec)]
valid = tab[:, 2] == tr._ws_tag                       # (round 5: the records of the last call; the table has holes)
rec, ir = rec[valid], tab[valid]
n_items = len(ir)
life = rec[:, :, :7].sum(axis=2).max(axis=1)          # a workgroup holds its slot until its last wave is done
sl, ns = ir[:, 1] & 0xffff, ir[:, 1] >> 16
def makespan(order, slots=2048):
    h = [0.0] * slots; heapq.heapify(h); end = 0.0
    for i in order:
        t = heapq.heappop(h) + life[i]; end = max(end, t); heapq.heappush(h, t)
    return end / 2400.0
idx = np.arange(n_items)
print(f"{name} spread={spread}: {n_items} workgroups, lifetime mean {life.mean():.0f} p95 {np.percentile(life,95):.0f} max {life.max():.0f} ticks; "
      f"sum / 2048 slots = {life.sum()/2048/2400:.1f} us, longest = {life.max()/2400:.1f} us")
print(f"  as dispatched                      {makespan(idx):6.1f} us")
print(f"  longest first (ignores the contract) {makespan(np.argsort(-life)):6.1f} us")
print(f"  single-slice tiles last            {makespan(np.concatenate([idx[ns > 1], idx[ns == 1]])):6.1f} us")
print(f"  slice-major (all tiles' slice 0, 1, ...) {makespan(np.lexsort((idx, sl))):6.1f} us")
for s_ in (0, 1, 4, 8, 16):
    m = sl >= s_
    if m.any():
        print(f"  slices >= {s_:2d}: {int(m.sum()):5d} workgroups, lifetime mean {life[m].mean():7.0f} ticks, head {rec[m][:, :, 0].mean():6.0f} walk {rec[m][:, :, 2].mean():6.0f} look-back {rec[m][:, :, 4].mean():6.0f}")
print(f"  ns == 1 : {int((ns == 1).sum())} workgroups, lifetime mean {life[ns == 1].mean():.0f}")
