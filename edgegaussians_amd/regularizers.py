"""Nearest neighbours + the two orientation regularisers on device.

Host-side mirror of `EdgeGaussianSplatting.k_nearest_sklearn` / `update_nearest_neighbors`
(edge_gs.py:135-151,326-344), `compute_direction_loss` (:346-373) and `compute_ratio_loss`
(:375-380).  The reference builds a CPU KD-tree over all means (D2H copy included) every 5th step of
the last 150 epochs; here the search is a uniform-grid kernel on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import torch
from torch import Tensor

from ._lib import call, load, ptr, stream


def make_grid(points: Tensor, margin: float = 0.0):
    """Uniform search grid over the points' bounding box (one host sync): (origin, cell edge, dims).  With
    `margin` > 0 the box is inflated by that fraction so that the grid can be REUSED while the points move: a
    point that leaves the box is clamped into a boundary cell, which keeps the search exact (every cell still
    lies at least as far from a query as its index distance says) and only costs speed."""
    pts = points.detach()
    N = pts.shape[0]
    lo_h, hi_h = pts.min(dim=0).values.tolist(), pts.max(dim=0).values.tolist()
    ext = [max(h - l, 1e-6) for l, h in zip(lo_h, hi_h)]
    lo_h = [l - margin * e for l, e in zip(lo_h, ext)]
    ext = [e * (1.0 + 2.0 * margin) for e in ext]
    vol = ext[0] * ext[1] * ext[2]
    # ~8 points per cell if they fill the box: the 27 cells around a query then hold its K <= 32 neighbours and the
    # first round settles it (a range look-up costs ~10 candidate evaluations: fewer, fuller cells win)
    cell = max((vol * 8.0 / max(N, 1)) ** (1.0 / 3.0), max(ext) / 512.0)
    lo_t = torch.tensor(lo_h, device=pts.device)

    def dims_of(c):
        return [max(1, min(512, int(math.ceil(e / c)))) for e in ext]

    # trained Gaussians sit on curves, not in the volume: shrink the cells until an OCCUPIED cell holds ~8 points
    # (a query thread walks every point of the 27 cells around it), within 2^24 cells of scratch
    for _ in range(4):
        d = dims_of(cell)
        ids = ((pts - lo_t) / cell).floor().long().clamp_(min=0)
        ids = (ids[:, 2].clamp_(max=d[2] - 1) * d[1] + ids[:, 1].clamp_(max=d[1] - 1)) * d[0] + ids[:, 0].clamp_(max=d[0] - 1)
        occ = N / max(int(torch.unique(ids).numel()), 1)
        if occ <= 16.0:
            break
        smaller = cell * max((8.0 / occ) ** 0.5, 0.25)
        ds = dims_of(smaller)
        if smaller < max(ext) / 512.0 or ds[0] * ds[1] * ds[2] > (1 << 24):
            break
        cell = smaller
    return lo_h, float(cell), dims_of(cell)


# up to this many points the exhaustive search (eg_knn_small: N^2 pairs over the whole chip, no grid, no host sync)
# beats the grid search on trained-like clouds (curves + floaters) and matches it on uniform ones: 0.10 vs 0.27-0.40 ms
# at 10 k points, 0.29 vs 0.30-0.51 ms at 20 k, 0.48 vs 0.27-0.53 ms at 32 k (tools/bench_regularizers.py)
KNN_EXHAUSTIVE_MAX = 24576


def knn(points: Tensor, k: int, want_dist: bool = False, grid=None, method: str = "auto", out: Tensor = None,
        scratch: Tensor = None) -> Tuple[Tensor, Tensor]:
    """Indices [N,k] (int32, ascending distance, self excluded) and, optionally, distances [N,k].
    method "auto": exhaustive search for N <= KNN_EXHAUSTIVE_MAX (no host sync at all), else the uniform-grid
    search (one host sync for the bounding box, `make_grid`, unless a grid is passed in) -- where the reference
    had a full D2H copy + CPU tree build.  "grid" / "exhaustive" force one.  `out` [N,k] int32 / `scratch` (int32,
    eg_knn_small_scratch_bytes): caller-owned buffers of the exhaustive search (a training loop reuses them)."""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3
    assert 1 <= k <= 32
    pts = points.detach().contiguous()
    N = pts.shape[0]
    dev = pts.device
    if method == "exhaustive" or (method == "auto" and N <= KNN_EXHAUSTIVE_MAX):
        idx = torch.empty(N, k, dtype=torch.int32, device=dev) if out is None else out
        assert idx.shape == (N, k) and idx.dtype == torch.int32 and idx.is_contiguous()
        d2 = torch.empty(N, k, device=dev) if want_dist else None
        nbytes = int(load().eg_knn_small_scratch_bytes(N, k))
        if scratch is None or scratch.numel() * 4 < nbytes:
            scratch = torch.empty(max(nbytes, 8) // 4, dtype=torch.int32, device=dev)
        call("eg_knn_small", ptr(pts), N, k, ptr(scratch), ptr(idx), ptr(d2) if d2 is not None else None, stream())
        return idx, (d2.sqrt() if d2 is not None else None)
    lo_h, cell, dims = grid if grid is not None else make_grid(pts)
    ncell = dims[0] * dims[1] * dims[2]
    cell_of = torch.empty(N, dtype=torch.int32, device=dev)
    counts = torch.zeros(ncell, dtype=torch.int32, device=dev)
    start = torch.empty(ncell + 1, dtype=torch.int32, device=dev)
    order = torch.empty(N, 4, device=dev)  # the points in cell order: x y z index
    idx = torch.empty(N, k, dtype=torch.int32, device=dev) if out is None else out
    d2 = torch.empty(N, k, device=dev) if want_dist else None
    origin = (C.c_float * 3)(*lo_h)
    cdims = (C.c_int32 * 3)(*dims)
    call("eg_knn", ptr(pts), N, k, origin, float(cell), cdims, ptr(cell_of), ptr(counts), ptr(start), ptr(order),
         ptr(idx), ptr(d2) if d2 is not None else None, stream())
    return idx, (d2.sqrt() if d2 is not None else None)


def reference_nn_indices(points: Tensor, dir_loss_num_nn: int, enforce_method: str = "enforce_full", grid=None,
                         method: str = "auto") -> Tensor:
    """`update_nearest_neighbors` (edge_gs.py:326-344): k_nearest_sklearn(points, k+1) -- 2k+1 for
    'enforce_half' -- already drops the point itself, and `indices[:, 1:]` then drops the NEAREST
    neighbour as well: the reference aligns with neighbours 2 .. k+1 (2 .. 2k+1).  Kept as is."""
    n = 2 * dir_loss_num_nn + 1 if enforce_method == "enforce_half" else dir_loss_num_nn + 1
    idx, _ = knn(points, n, grid=grid, method=method)
    return idx[:, 1:].contiguous()


def direction_loss(means: Tensor, quats: Tensor, log_scales: Tensor, nn_idx: Tensor, top_k: int = 0):
    """Returns (loss [device scalar], dloss/dmeans [N,3], dloss/dquats [N,4]) of edge_gs.py:346-373.
    top_k = dir_loss_num_nn with a [N,2k] neighbour table is the 'enforce_half' method (:366-369)."""
    N, K = nn_idx.shape
    g_means = torch.zeros(N, 3, device=means.device)
    g_quats = torch.empty(N, 4, device=means.device)
    s = torch.zeros(1, device=means.device)
    call("eg_direction_loss", ptr(means.contiguous()), ptr(quats.contiguous()), ptr(log_scales.contiguous()),
         ptr(nn_idx.contiguous()), N, K, int(top_k), ptr(g_means), ptr(g_quats), ptr(s), stream())
    w = -1.0 / (N * (top_k if 0 < top_k < K else K))
    return 1.0 + w * s[0], g_means * w, g_quats * w


def ratio_loss(log_scales: Tensor):
    """Returns (loss [device scalar], dloss/dlog_scales [N,3]) of edge_gs.py:375-380."""
    N = log_scales.shape[0]
    g = torch.empty(N, 3, device=log_scales.device)
    s = torch.zeros(1, device=log_scales.device)
    call("eg_ratio_loss", ptr(log_scales.contiguous()), N, ptr(g), ptr(s), stream())
    return s[0] / N, g / N
