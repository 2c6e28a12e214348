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
    """Uniform search grid over the points' bounding box chosen on the HOST (one sync): (origin, cell edge, dims),
    ~4 points per cell if they fill the box.  Only for callers that want to fix the grid themselves (`knn(...,
    grid=...)`, the C entry eg_knn); `knn` by default lets the device choose (eg_knn_auto: no sync).  With
    `margin` > 0 the box is inflated by that fraction so that the grid can be REUSED while the points move: a point
    that leaves the box is clamped into a boundary cell, which keeps the search exact (every cell still lies at
    least as far from a query as its index distance says) and only costs speed."""
    pts = points.detach()
    N = pts.shape[0]
    lo_h, hi_h = pts.min(dim=0).values.tolist(), pts.max(dim=0).values.tolist()
    ext = [max(h - l, 1e-6) for l, h in zip(lo_h, hi_h)]
    lo_h = [l - margin * e for l, e in zip(lo_h, ext)]
    ext = [e * (1.0 + 2.0 * margin) for e in ext]
    vol = ext[0] * ext[1] * ext[2]
    cell = max((vol * 4.0 / max(N, 1)) ** (1.0 / 3.0), max(ext) / 256.0)
    dims = [max(1, min(256, int(math.ceil(e / cell)))) for e in ext]
    return lo_h, float(cell), dims


# up to this many points the exhaustive search (eg_knn_small: one launch, N^2 pairs over the whole chip) beats the grid
# search (six small launches): 35 us against 43-54 us at 4 k points; at 10 k it is 105 against 50-64
# (tools/bench_regularizers.py)
KNN_EXHAUSTIVE_MAX = 5000
_knn_scratch = {}


def _grid_buffers(N: int, ncell: int, dev):
    """cell_of [N], counts [ncell] (zero between calls), start [ncell + 1], order [N,4], 64 bytes of grid scratch --
    cached per (N, ncell, device): a training loop calls this every few steps with the same sizes."""
    key = (N, ncell, str(dev))
    b = _knn_scratch.get(key)
    if b is None:
        if len(_knn_scratch) > 8:
            _knn_scratch.clear()
        b = (torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(ncell, dtype=torch.int32, device=dev),
             torch.empty(ncell + 1, dtype=torch.int32, device=dev), torch.empty(N, 4, device=dev),
             torch.zeros(16, dtype=torch.int32, device=dev))
        _knn_scratch[key] = b
    return b


def knn(points: Tensor, k: int, want_dist: bool = False, grid=None, method: str = "auto",
        out: Tensor = None, kth: Tensor = None, kth_slack: float = 1.2) -> Tuple[Tensor, Tensor]:
    """Indices [N,k] (int32, ascending distance, self excluded) and, optionally, distances [N,k] -- exact, and
    without a host sync: method "auto" = exhaustive search for N <= KNN_EXHAUSTIVE_MAX (eg_knn_small), else the
    uniform-grid search on a grid the DEVICE chooses (eg_knn_auto: bounding box by a reduction kernel, cell count a
    function of N) -- where the reference had a full D2H copy + CPU tree build.  "exhaustive" / "grid" force one;
    `grid` = a host-chosen grid from `make_grid` (eg_knn).  `out` [N,k] int32: caller-owned result buffer.
    `kth` [N] float32 (device-grid search only): every point's K-th squared distance of the caller's PREVIOUS search
    of the same, slowly moving points (zeros = unknown); read as the entry bound of the point's list (times
    kth_slack), overwritten with this search's -- ~K insertions per query instead of ~K ln(n / K); the result does
    not depend on it."""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3
    assert 1 <= k <= 32
    pts = points.detach().contiguous()
    N = pts.shape[0]
    dev = pts.device
    idx = torch.empty(N, k, dtype=torch.int32, device=dev) if out is None else out
    assert idx.shape == (N, k) and idx.dtype == torch.int32 and idx.is_contiguous()
    d2 = torch.empty(N, k, device=dev) if want_dist else None
    d2p = ptr(d2) if d2 is not None else None
    if method == "exhaustive" or (method == "auto" and grid is None and N <= KNN_EXHAUSTIVE_MAX):
        call("eg_knn_small", ptr(pts), N, k, ptr(idx), d2p, stream())
    elif grid is None:
        D = int(load().eg_knn_auto_dims(N, k))
        cell_of, counts, start, order, gs = _grid_buffers(N, D * D * D, dev)
        if kth is not None:
            assert kth.shape == (N,) and kth.dtype == torch.float32 and kth.is_cuda and kth.is_contiguous()
        call("eg_knn_auto", ptr(pts), N, k, ptr(cell_of), ptr(counts), ptr(start), ptr(order), ptr(gs), ptr(idx), d2p,
             ptr(kth) if kth is not None else None, float(kth_slack), stream())
    else:
        lo_h, cell, dims = grid
        cell_of, counts, start, order, _ = _grid_buffers(N, dims[0] * dims[1] * dims[2], dev)
        origin = (C.c_float * 3)(*lo_h)
        cdims = (C.c_int32 * 3)(*dims)
        call("eg_knn", ptr(pts), N, k, origin, float(cell), cdims, ptr(cell_of), ptr(counts), ptr(start), ptr(order),
             ptr(idx), d2p, stream())
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
