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
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

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
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100_000)
