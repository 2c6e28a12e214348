"""Cost of a LATE-epoch iteration of the real loop (train_loop.train_epoch with the orientation regularisers on: kNN +
direction loss + ratio loss + their Adam steps after every 5th projection step) next to a projection-only epoch, on the
trained-like variants of config 1 / 2 at the reference's full learning rates.  Needs an MI355X."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edgegaussians_amd import train_loop
cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "abc_train_config.json")))
proj, orient = cfg["training"]["loss"]["projection_losses"], cfg["training"]["loss"]["orientation_losses"]
for name in ("config1", "config2"):
    tr, sc, whole, ratio, poses = bench.build_trainer(name, 0, "cuda:0", spread_opacity=True)
    tr.ensure_capacity()
    n_views = sc.viewmats.shape[0]
    for late in (False, True):
        o = dict(orient)
        o["start_dir_loss_at_epoch"] = -1 if late else 10**6
        o["start_ratio_loss_at_epoch"] = -1 if late else 10**6
        for _ in range(3):
            train_loop.train_epoch(tr, list(range(n_views)), 300, 400, proj, o)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E = 6
        for e in range(E):
            train_loop.train_epoch(tr, list(range(n_views)), 300, 400, proj, o, read_back=(e == E - 1))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, "late-epoch regularisers" if late else "projection only", round(1e6 * dt / (E * n_views), 1), "us/step", flush=True)
