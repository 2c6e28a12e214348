#!/usr/bin/env python3
"""Timing of the 8(f) rank-1 row: kNN + direction/ratio regulariser step, GPU vs the reference's CPU
KD-tree (sklearn) on the same host.  Prints one JSON line."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from edgegaussians_amd import regularizers as R  # noqa: E402
from edgegaussians_amd import synth  # noqa: E402


def main(n=100_000):
    sc = synth.make_scene(n, 1, 64, 64, seed=0)
    pts = sc.means.cuda()
    q, ls = sc.quats.cuda(), sc.log_scales.cuda()

    def timed(fn, reps):
        """median of `reps` synchronised calls, ms (a first call after an allocation can take milliseconds)"""
        fn()
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return sorted(ts)[len(ts) // 2]

    nn = R.reference_nn_indices(pts, 5)
    out = {"n": n,
           "knn_k6_ms": timed(lambda: R.reference_nn_indices(pts, 5), 20),
           "direction_loss_ms": timed(lambda: R.direction_loss(pts, q, ls, nn), 50),
           "ratio_loss_ms": timed(lambda: R.ratio_loss(ls), 50)}
    from sklearn.neighbors import NearestNeighbors
    x = sc.means.numpy()
    t0 = time.perf_counter()
    NearestNeighbors(n_neighbors=7, algorithm="auto", metric="euclidean").fit(x).kneighbors(x)
    out["sklearn_kdtree_cpu_ms"] = 1e3 * (time.perf_counter() - t0)
    # the two searches around the switch-over (regularizers.KNN_EXHAUSTIVE_MAX) and at scale (grid chosen on the
    # device, eg_knn_auto; K = 6 as ABC / DTU, K = 11 as Replica's dir_loss_num_nn = 10), uniform points and a trained-like
    # cloud (a quarter of the points on 12 line segments, the rest faint floaters through the volume), rows in
    # Morton order like EdgeTrainer(spatial_order=True) keeps them
    from edgegaussians_amd.trainer import EdgeTrainer  # noqa: F401  (Morton helper lives there)
    rows = []
    g = torch.Generator().manual_seed(0)
    for m in (2000, 4000, 10000, 32768, 100000, 500000):
        for kind in ("uniform", "trained-like"):
            p = torch.rand(m, 3, generator=g)
            if kind == "trained-like":
                t = torch.rand(m, 1, generator=g)
                seg = torch.randint(0, 12, (m,), generator=g)
                a, b = torch.rand(12, 3, generator=g), torch.rand(12, 3, generator=g)
                p[: m // 4] = (a[seg] * (1 - t) + b[seg] * t + 0.003 * torch.randn(m, 3, generator=g))[: m // 4]
            p = p.cuda()
            q16 = ((p - p.min(0).values) / (p.max(0).values - p.min(0).values + 1e-9) * 1023).long()
            code = torch.zeros(m, dtype=torch.long, device=p.device)
            for bit in range(10):
                for ax in range(3):
                    code |= ((q16[:, ax] >> bit) & 1) << (3 * bit + ax)
            p = p[torch.argsort(code)].contiguous()
            rows.append({"n": m, "points": kind,
                         "exhaustive_ms": timed(lambda: R.knn(p, 6, method="exhaustive"), 10) if m <= 65536 else None,
                         "grid_ms": timed(lambda: R.knn(p, 6, method="grid"), 10),
                         "grid_k11_ms": timed(lambda: R.knn(p, 11, method="grid"), 10)})
    out["knn_k6_by_size"] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100_000)
