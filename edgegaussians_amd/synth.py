"""Seeded synthetic workloads of the shapes BASELINE.json names.

Gaussians: uniform in the reference's init box (configs/ABC_DexiNed.json:28-29 ->
data_utils.py:72-75), constant log-scale log(0.004) (configs:33, edge_gs.py:80-81) with an
optional "trained-like" 5:1 major axis, random unit quaternions (misc_utils.py:36-51 formula),
logit-opacity logit(0.08) (configs:35, edge_gs.py:93) or a spread variant.

Cameras: synthetic look-at poses on a sphere around the box (the real poses of scan 00004926
are a test fixture under tests/golden and are used when a path to them is given).  GT edge
maps: the projected wireframe of the unit cube, ~0.7 % edge density like the DexiNed maps.
Everything is generated on the CPU with a private torch.Generator, so a (seed, N, V, H, W)
tuple names one workload on every machine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


@dataclass
class Scene:
    means: torch.Tensor  # [N,3]
    log_scales: torch.Tensor  # [N,3]
    quats: torch.Tensor  # [N,4] wxyz, un-normalised
    logit_opacities: torch.Tensor  # [N,1]
    viewmats: torch.Tensor  # [V,4,4] world->cam
    Ks: torch.Tensor  # [V,3,3]
    gt: torch.Tensor  # [V,H,W] in [0,1]
    width: int
    height: int


def random_quats(n: int, g: torch.Generator) -> torch.Tensor:
    u, v, w = torch.rand(n, generator=g), torch.rand(n, generator=g), torch.rand(n, generator=g)
    return torch.stack([
        torch.sqrt(1 - u) * torch.sin(2 * math.pi * v),
        torch.sqrt(1 - u) * torch.cos(2 * math.pi * v),
        torch.sqrt(u) * torch.sin(2 * math.pi * w),
        torch.sqrt(u) * torch.cos(2 * math.pi * w)], dim=-1)


def lookat_cameras(n_views: int, width: int, height: int, g: torch.Generator,
                   radius: float = 3.9, focal_over_width: float = 1111.11 / 800.0):
    """Poses on a sphere of the scan's camera distance (|c| ~ 3.9 in meta_data.json) looking at
    the box centre; OpenCV convention (x right, y down, z forward), as cameras.py:103-127."""
    centre = torch.tensor([0.5, 0.5, 0.5])
    vms, Ks = [], []
    for i in range(n_views):
        phi = 2 * math.pi * (i + 0.37) / n_views
        theta = math.radians(35.0 + 40.0 * float(torch.rand((), generator=g)))
        c = centre + radius * torch.tensor([math.cos(phi) * math.sin(theta),
                                            math.sin(phi) * math.sin(theta), math.cos(theta)])
        z = centre - c
        z = z / z.norm()
        up = torch.tensor([0.0, 0.0, 1.0])
        x = torch.linalg.cross(z, up)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        R = torch.stack([x, y, z])  # world->cam rows
        t = -R @ c
        vm = torch.eye(4)
        vm[:3, :3] = R
        vm[:3, 3] = t
        f = focal_over_width * width
        K = torch.tensor([[f, 0, (width - 1) / 2.0], [0, f, (height - 1) / 2.0], [0, 0, 1.0]])
        vms.append(vm)
        Ks.append(K)
    return torch.stack(vms), torch.stack(Ks)


def wireframe_edge_maps(viewmats, Ks, width, height) -> torch.Tensor:
    """Rasterises the 12 edges of the unit cube plus two inner loops as 1-px anti-aliased lines."""
    corners = torch.tensor([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)]) * 0.7 + 0.15
    edges = [(a, b) for a in range(8) for b in range(a + 1, 8)
             if int((corners[a] != corners[b]).sum()) == 1]
    segs = [(corners[a], corners[b]) for a, b in edges]
    ts = torch.linspace(0, 1, 33)
    ring = torch.stack([0.5 + 0.22 * torch.cos(2 * math.pi * ts), 0.5 + 0.22 * torch.sin(2 * math.pi * ts),
                        torch.full_like(ts, 0.85)], dim=-1)
    segs += [(ring[i], ring[i + 1]) for i in range(32)]
    out = torch.zeros(viewmats.shape[0], height, width)
    for v in range(viewmats.shape[0]):
        R, t, K = viewmats[v, :3, :3], viewmats[v, :3, 3], Ks[v]
        img = out[v]
        for p0, p1 in segs:
            n = 4 * max(width, height)
            s = torch.linspace(0, 1, n)[:, None]
            P = (p0[None] * (1 - s) + p1[None] * s) @ R.T + t
            uv = P[:, :2] / P[:, 2:3]
            px = (K[0, 0] * uv[:, 0] + K[0, 2]).round().long()
            py = (K[1, 1] * uv[:, 1] + K[1, 2]).round().long()
            ok = (px >= 0) & (px < width) & (py >= 0) & (py < height) & (P[:, 2] > 0)
            img[py[ok], px[ok]] = 1.0
    return out


def make_scene(n_gauss: int, n_views: int, width: int, height: int, seed: int = 0,
               anisotropy: float = 5.0, spread_opacity: bool = False,
               cameras_npz: Optional[str] = None, scale: float = 0.004) -> Scene:
    g = torch.Generator().manual_seed(seed)
    means = 1.1 * torch.rand(n_gauss, 3, generator=g) - 0.55 + 0.5
    log_scales = torch.full((n_gauss, 3), math.log(scale))
    if anisotropy != 1.0:
        log_scales[:, 0] += math.log(anisotropy)
    quats = random_quats(n_gauss, g)
    if spread_opacity:
        op = 0.05 + 0.85 * torch.rand(n_gauss, 1, generator=g)
    else:
        op = torch.full((n_gauss, 1), 0.08)
    logit = torch.logit(op)
    if cameras_npz is not None:
        # real poses (the reference's scan, tests/golden/cameras_00004926.npz), intrinsics rescaled to the
        # requested size.  The fixture's principal point is (W - 1) / 2 = 399.5 in pixel-INDEX coordinates, so
        # it is rescaled about the pixel-centre convention: c' = (c + 0.5) s - 0.5  (800 -> 512: 255.5)
        d = np.load(cameras_npz)
        sx, sy = width / float(d["width"]), height / float(d["height"])
        Ks = torch.from_numpy(d["Ks"]).clone()[:n_views]
        Ks[:, 0, 0] *= sx
        Ks[:, 1, 1] *= sy
        Ks[:, 0, 2] = (Ks[:, 0, 2] + 0.5) * sx - 0.5
        Ks[:, 1, 2] = (Ks[:, 1, 2] + 0.5) * sy - 0.5
        vms = torch.from_numpy(d["viewmats"]).clone()[:n_views]
        assert vms.shape[0] == n_views, f"the fixture holds {vms.shape[0]} poses, {n_views} were asked for"
    else:
        vms, Ks = lookat_cameras(n_views, width, height, g)
    gt = wireframe_edge_maps(vms, Ks, width, height)
    return Scene(means, log_scales, quats, logit, vms, Ks, gt, width, height)


def weight_map(strategy: str, gt: torch.Tensor, ratio: float = 1.0,
               generator: Optional[torch.Generator] = None, threshold: float = 0.5) -> torch.Tensor:
    """Per-pixel weight map w such that the reference's projection loss (edge_gs.py:288-324) is
    sum_p w_p * |render_p - gt_p|.  `gt` is [H,W]; the edge mask is gt >= 0.5 (edge_gs.py:49,158).
    The bg_edge_ratio sample is drawn here, on the host, like the reference does (CPU randperm,
    edge_gs.py:306) -- the weight map is an INPUT of the device step."""
    edge = gt >= threshold
    hw = edge.numel()
    H, W = edge.shape
    if strategy == "whole":
        return torch.full((H, W), 1.0 / hw)
    if strategy == "weighted":
        n_e = int(edge.sum())
        n_b = hw - n_e
        w = torch.where(edge, torch.tensor(n_b / hw), torch.tensor(n_e / hw)).float()
        return w / hw
    if strategy == "bg_edge_ratio":
        n_e = int(edge.sum())
        n_sel = int(ratio * n_e)
        perm = torch.randperm(hw - n_e, generator=generator)[:n_sel] % hw
        sel = torch.zeros(hw, dtype=torch.bool)
        sel[perm] = True
        w = edge.reshape(-1).float() / max(n_e, 1) + sel.float() / max(int(sel.sum()), 1)
        return w.reshape(H, W)
    raise ValueError(f"Unknown projection loss strategy: {strategy}")
