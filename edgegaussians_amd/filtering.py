"""Post-hoc Gaussian filters of the edge-extraction stage, on device.

Host-side mirror of `/root/reference/edgegaussians/edge_extraction/filtering.py`:
    filter_by_projection   :80-123   -> eg_project_visibility (one N x V kernel instead of a V-iteration
                                        numpy loop with a Python list round trip per view)
    filter_by_opacity      :71-77    -> one comparison
`filter_stat_outliers` (:59-69) is Open3D's statistical outlier removal; it is outside SURVEY 8 and is
not provided here.

Same argument meaning and return value as the reference: a boolean inlier mask of shape [N].
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

from ._lib import call, ptr, stream


def pack_cameras(cameras: Sequence[Dict], device) -> torch.Tensor:
    """[V,21] = K (9) | R (9) | t (3) from the reference's camera dicts (filtering.py:42-56)."""
    rows = [np.concatenate([np.asarray(c["K"], np.float32).reshape(9), np.asarray(c["R"], np.float32).reshape(9),
                            np.asarray(c["t"], np.float32).reshape(3)]) for c in cameras]
    return torch.from_numpy(np.stack(rows)).to(device).contiguous()


def filter_by_projection(gaussian_means, edge_images, cameras: Sequence[Dict], visib_thresh: float = 0.1,
                         device="cuda") -> np.ndarray:
    """filtering.py:80-123: keep Gaussians whose mean, projected into every view and rounded to a
    pixel, sees an average edge strength above `visib_thresh` (views it falls outside of count as 0)."""
    means = torch.as_tensor(np.asarray(gaussian_means, np.float32)).to(device).contiguous()
    V = len(edge_images)
    if V == 0 or means.shape[0] == 0:
        return np.zeros(means.shape[0], dtype=bool)
    h, w = int(cameras[0]["h"]), int(cameras[0]["w"])
    maps = torch.stack([torch.as_tensor(e).to(device=device, dtype=torch.float32) for e in edge_images]).contiguous()
    if tuple(maps.shape) != (V, h, w):
        raise ValueError(f"edge_images must be {V} maps of {h}x{w}, got {tuple(maps.shape)}")
    visib = torch.zeros(means.shape[0], dtype=torch.float64, device=device)  # numpy's accumulator type there
    call("eg_project_visibility", ptr(means), means.shape[0], ptr(pack_cameras(cameras, device)), V, ptr(maps), w, h,
         ptr(visib), stream())
    return (visib.cpu().numpy() / float(V) > visib_thresh).reshape(-1)


def filter_by_opacity(opacities, min_opacity: float) -> np.ndarray:
    """filtering.py:71-77."""
    return (np.asarray(opacities) > min_opacity).reshape(-1)
