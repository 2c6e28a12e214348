#!/usr/bin/env python3
"""round 6: the step on a CU-MASKED device.  The forward's look-back relies on workgroups being dispatched
in index order (a slice only ever waits for lower-indexed workgroups) and places its records by `workgroup b -> XCD b % 8` for
locality.  This script trains a few steps of a stop-heavy scene in THIS process -- run it once plain and once under
HSA_CU_MASK / ROC_GLOBAL_CU_MASK -- and writes the resulting state to an .npz: a masked device must give the same parameters
(locality may go, correctness may not) and must not trip the bounded look-back poll.
usage: python tools/cu_mask_check.py out.npz"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edgegaussians_amd import EdgeTrainer, LRSchedule, synth  # noqa: E402

sc = synth.make_scene(20000, 6, 512, 512, seed=5, anisotropy=5.0, spread_opacity=True, scale=0.008)
sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, 512, 512, schedule=sched)
tr.ensure_capacity()
w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(6)]
views = [0, 3, 1, 5, 2, 4] * 4
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.train_steps(views, [w[v] for v in views])
loss = tr.pop_loss()
dt = time.perf_counter() - t0
sd = {k: v.cpu().numpy() for k, v in tr.state_dict().items()}
np.savez(sys.argv[1], loss=np.float64(loss), stall=np.int32(any(tr._ctl_bits())), max_tile=np.int32(tr.max_tile_seen),
         rewalk_hint=np.int32(tr.rewalk_hint), **sd)
print(f"mask env: HSA_CU_MASK={os.environ.get('HSA_CU_MASK')} ROC_GLOBAL_CU_MASK={os.environ.get('ROC_GLOBAL_CU_MASK')}; "
      f"CUs reported {torch.cuda.get_device_properties(0).multi_processor_count}; 24 steps in {1e3 * dt:.2f} ms, loss {loss:.6f}, "
      f"largest tile {tr.max_tile_seen}, stops seen {tr.rewalk_hint}, replays {tr.overflow_events + tr.rewalk_misses}")
