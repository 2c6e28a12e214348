"""CPU oracle for the edge-Gaussian rasterizer hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``edgegaussians_amd``) never does, and fails loudly when its
HIP library is missing.

What it restates
----------------
The single call ``gsplat.rasterization(...)`` made at
``/root/reference/edgegaussians/models/edge_gs.py:250-268`` plus the per-step glue around it
(``edge_gs.py:278-324,603-613``, ``models/losses.py:5-11``, ``train_gaussians.py:83-106``).
The arithmetic of that call lives in the third-party dependency ``gsplat==1.0.0``
(``/root/reference/requirements.txt:64``), which is NOT vendored under ``/root/reference`` and is
not installed in this image.  Its published algorithm (gsplat 1.0.0: ``rendering.py::rasterization``,
``cuda/_torch_impl.py``, ``cuda/csrc/{fully_fused_projection,isect_tiles,rasterize_to_pixels}_*.cu``)
is restated here from the description in SURVEY.md section 2.3 / 8a as *dense PyTorch*, with
autograd supplying every backward pass, so the oracle's gradients are derived independently of
the hand-written HIP backward kernels.

PARITY UNPINNED for the rasterizer arithmetic: the reference holds no test, golden vector or
fixture for this path and gsplat cannot be run here.  The pieces of the path that DO live in
the reference tree (losses, weight masks, quaternion convention, cameras, LR schedule, densify/
cull) are pinned by the fixtures under ``tests/golden`` generated from the importable reference
modules (``tests/golden/make_golden.py``).  The rasterizer constants are pinned by closed-form
known-answer tests and float64 ``gradcheck`` in ``tests/test_oracle.py``.

Conventions (all from gsplat 1.0.0 as used by the reference's arguments):
  * quaternions are (w, x, y, z), normalised inside;  Sigma = (R S)(R S)^T
  * pixel centres at (j + 0.5, i + 0.5); 16x16 tiles; sort key = (tile_id << 32) | float_bits(depth)
  * antialiased mode: Sigma2D += 0.3 I, opacity *= sqrt(max(0, det0/det1))
  * alpha = min(0.999, o * exp(-sigma)); skip if sigma < 0 or alpha < 1/255; stop BEFORE the
    Gaussian that would take transmittance to <= 1e-4
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

EPS2D = 0.3
ALPHA_MAX = 0.999
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4
FOV_CLAMP = 1.3
RADIUS_SIGMAS = 3.0      # radius = ceil(3 sqrt(lambda_max))
RADIUS_DET_FLOOR = 0.01  # lambda_max = b + sqrt(max(0.01, b^2 - det))
PIXEL_CENTRE = 0.5       # pixel (i, j) is sampled at (j + 0.5, i + 0.5)
STOP_BEFORE = True       # the walk ends BEFORE the Gaussian that would take T to <= T_STOP (it does not contribute)
# (every restated constant is a module-level name so that tests/test_oracle_mutations.py can flip it and watch a
# known-answer test of tests/test_oracle.py fail)


# --------------------------------------------------------------------------------------
# G1: projection
# --------------------------------------------------------------------------------------
def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> R [N,3,3]; normalisation by rsqrt of the squared norm (no eps clamp).

    Same convention as the reference's own helper ``misc_utils.py:53-86`` (checked in
    tests/test_golden.py against fixtures generated from it)."""
    inv = torch.rsqrt((quats * quats).sum(-1, keepdim=True))
    q = quats * inv
    w, x, y, z = q.unbind(-1)
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    wx, wy, wz = w * x, w * y, w * z
    R = torch.stack(
        [
            1 - 2 * (y2 + z2), 2 * (xy - wz), 2 * (xz + wy),
            2 * (xy + wz), 1 - 2 * (x2 + z2), 2 * (yz - wx),
            2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (x2 + y2),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]  # scale the columns
    return M @ M.transpose(-1, -2)


class _Compensation(torch.autograd.Function):
    """comp = sqrt(max(0, det0/det1)) with gsplat's backward 0.5 * v / (comp + 1e-6)."""

    @staticmethod
    def forward(ctx, ratio):
        comp = torch.sqrt(torch.clamp(ratio, min=0.0))
        ctx.save_for_backward(comp, ratio)
        return comp

    @staticmethod
    def backward(ctx, v):
        comp, ratio = ctx.saved_tensors
        g = 0.5 * v / (comp + 1e-6)
        return torch.where(ratio >= 0, g, torch.zeros_like(g))


def project(
    means: torch.Tensor,  # [N,3]
    quats: torch.Tensor,  # [N,4] wxyz
    scales: torch.Tensor,  # [N,3] (already exp'ed)
    viewmat: torch.Tensor,  # [4,4] world->cam
    K: torch.Tensor,  # [3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    eps2d: float = EPS2D,
    radius_clip: float = 0.0,
):
    """Returns radii i32[N], means2d[N,2], depths[N], conics[N,3], compensations[N].

    Culled Gaussians get radius 0 and zeros in every float output (gsplat leaves them
    uninitialised; nothing downstream reads them)."""
    dt = means.dtype
    Rv = viewmat[:3, :3].to(dt)
    tv = viewmat[:3, 3].to(dt)
    fx, fy, cx, cy = K[0, 0].to(dt), K[1, 1].to(dt), K[0, 2].to(dt), K[1, 2].to(dt)

    mean_c = means @ Rv.T + tv  # [N,3]
    x, y, z = mean_c.unbind(-1)
    in_z = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(in_z, z, torch.ones_like(z))  # keep culled rows finite

    covar = quat_scale_to_covar(quats, scales)
    covar_c = Rv @ covar @ Rv.T

    lim_x = FOV_CLAMP * (0.5 * width / fx)
    lim_y = FOV_CLAMP * (0.5 * height / fy)
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_x, torch.maximum(-lim_x, x * rz))
    ty = zs * torch.minimum(lim_y, torch.maximum(-lim_y, y * rz))
    zero = torch.zeros_like(rz)
    J = torch.stack(
        [fx * rz, zero, -fx * tx * rz2, zero, fy * rz, -fy * ty * rz2], dim=-1
    ).reshape(-1, 2, 3)
    cov2d = J @ covar_c @ J.transpose(-1, -2)  # [N,2,2]
    mean2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    c00, c01, c11 = cov2d[:, 0, 0], cov2d[:, 0, 1], cov2d[:, 1, 1]
    det0 = c00 * c11 - c01 * cov2d[:, 1, 0]
    b00 = c00 + eps2d
    b11 = c11 + eps2d
    det1 = b00 * b11 - c01 * cov2d[:, 1, 0]
    det_ok = det1 > 0
    det1s = torch.where(det_ok, det1, torch.ones_like(det1))
    comp = _Compensation.apply(det0 / det1s)

    inv = 1.0 / det1s
    conic = torch.stack([b11 * inv, -c01 * inv, b00 * inv], dim=-1)

    with torch.no_grad():
        bh = 0.5 * (b00 + b11)
        v1 = bh + torch.sqrt(torch.clamp(bh * bh - det1, min=RADIUS_DET_FLOOR))
        radius = torch.ceil(RADIUS_SIGMAS * torch.sqrt(v1))
        ok = in_z & det_ok & (radius > radius_clip)
        ok &= ~(
            (mean2d[:, 0] + radius <= 0)
            | (mean2d[:, 0] - radius >= width)
            | (mean2d[:, 1] + radius <= 0)
            | (mean2d[:, 1] - radius >= height)
        )
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    mean2d = torch.where(ok[:, None], mean2d, torch.zeros_like(mean2d))
    depths = torch.where(ok, z, torch.zeros_like(z))
    conic = torch.where(ok[:, None], conic, torch.zeros_like(conic))
    comp = torch.where(ok, comp, torch.zeros_like(comp))
    return radii, mean2d, depths, conic, comp


# --------------------------------------------------------------------------------------
# G2-G6: tile intersection, key build, stable sort, offsets -- integer work
# --------------------------------------------------------------------------------------
def tile_bounds(means2d: np.ndarray, radii: np.ndarray, tile: int, tw: int, th: int):
    """[lo, hi) tile box per Gaussian, float32 arithmetic exactly as the float pipeline does it:
    (x / tile) -/+ (r / tile) in fp32, floor / ceil, clamp to [0, tw] x [0, th]."""
    m = means2d.astype(np.float32)
    r = radii.astype(np.float32)
    ts = np.float32(tile)
    tr = r / ts
    txc = m[:, 0] / ts
    tyc = m[:, 1] / ts
    x0 = np.clip(np.floor(txc - tr), 0, tw).astype(np.int64)
    y0 = np.clip(np.floor(tyc - tr), 0, th).astype(np.int64)
    x1 = np.clip(np.ceil(txc + tr), 0, tw).astype(np.int64)
    y1 = np.clip(np.ceil(tyc + tr), 0, th).astype(np.int64)
    dead = radii <= 0
    x0[dead] = x1[dead] = y0[dead] = y1[dead] = 0
    return x0, y0, x1, y1


def isect_tiles(means2d, radii, depths, tile: int, tw: int, th: int):
    """Returns tiles_per_gauss i32[N], isect_ids i64[M], flatten_ids i32[M] (sorted, stable).

    key = (tile_id << 32) | raw bits of the positive fp32 depth; emission row-major over the
    tile box in Gaussian order; stable sort on the key (single camera => no camera bits)."""
    means2d = np.asarray(means2d, dtype=np.float32)
    radii = np.asarray(radii, dtype=np.int32)
    depths = np.asarray(depths, dtype=np.float32)
    x0, y0, x1, y1 = tile_bounds(means2d, radii, tile, tw, th)
    tpg = ((y1 - y0) * (x1 - x0)).astype(np.int32)
    M = int(tpg.sum())
    ids = np.empty(M, dtype=np.int64)
    flat = np.empty(M, dtype=np.int32)
    dbits = depths.view(np.int32).astype(np.int64)
    cur = 0
    for g in np.nonzero(tpg)[0]:
        ys = np.arange(y0[g], y1[g], dtype=np.int64)
        xs = np.arange(x0[g], x1[g], dtype=np.int64)
        t = (ys[:, None] * tw + xs[None, :]).reshape(-1)
        n = t.size
        ids[cur:cur + n] = (t << 32) | dbits[g]
        flat[cur:cur + n] = g
        cur += n
    order = np.argsort(ids, kind="stable")
    return tpg, ids[order], flat[order]


def isect_offset_encode(isect_ids: np.ndarray, tw: int, th: int) -> np.ndarray:
    """offsets[t] = first index of tile t's run in the sorted list (== #isects with tile < t)."""
    tiles = (np.asarray(isect_ids, dtype=np.int64) >> 32).astype(np.int64)
    return np.searchsorted(tiles, np.arange(tw * th, dtype=np.int64), side="left").astype(
        np.int32
    ).reshape(th, tw)


# --------------------------------------------------------------------------------------
# G7/G8: per-pixel front-to-back compositing; autograd is the backward
# --------------------------------------------------------------------------------------
def composite(
    means2d: torch.Tensor,  # [N,2]
    conics: torch.Tensor,  # [N,3]
    colors: torch.Tensor,  # [N,D]
    opacities: torch.Tensor,  # [N]
    width: int,
    height: int,
    tile: int,
    offsets: np.ndarray,  # [th,tw] i32
    flatten_ids: np.ndarray,  # [M] i32
    absgrad_buf: Optional[torch.Tensor] = None,  # [N,2], accumulated in backward hooks
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns render[H,W,D], alphas[H,W,1], last_ids i32[H,W]."""
    dt = means2d.dtype
    D = colors.shape[-1]
    th, tw = offsets.shape
    M = int(flatten_ids.shape[0])
    flat_off = offsets.reshape(-1)
    render = torch.zeros(height, width, D, dtype=dt)
    alphas = torch.zeros(height, width, 1, dtype=dt)
    last_ids = torch.zeros(height, width, dtype=torch.int32)
    out_r, out_a = [], []
    fid = torch.from_numpy(np.ascontiguousarray(flatten_ids)).long()

    for t in range(th * tw):
        s = int(flat_off[t])
        e = int(flat_off[t + 1]) if t + 1 < th * tw else M
        if e <= s:
            continue
        ti, tj = divmod(t, tw)
        i0, j0 = ti * tile, tj * tile
        i1, j1 = min(i0 + tile, height), min(j0 + tile, width)
        ii, jj = torch.meshgrid(
            torch.arange(i0, i1), torch.arange(j0, j1), indexing="ij"
        )
        py = ii.reshape(-1).to(dt) + PIXEL_CENTRE
        px = jj.reshape(-1).to(dt) + PIXEL_CENTRE
        g = fid[s:e]
        xy = means2d[g]
        con = conics[g]
        op = opacities[g]
        col = colors[g]
        dx = xy[None, :, 0] - px[:, None]
        dy = xy[None, :, 1] - py[:, None]
        a, b, c = con[None, :, 0], con[None, :, 1], con[None, :, 2]
        sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
        vis = torch.exp(-sigma)
        alpha_raw = op[None, :] * vis
        alpha = torch.clamp(alpha_raw, max=ALPHA_MAX)
        with torch.no_grad():
            valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
            a_m = torch.where(valid, alpha, torch.zeros_like(alpha))
            next_T = torch.cumprod(1 - a_m, dim=1)
            if not STOP_BEFORE:  # (mutation leg only: the Gaussian that crosses the threshold still contributes)
                next_T = torch.cat([torch.ones_like(next_T[:, :1]), next_T[:, :-1]], dim=1)
            stopped = torch.cummax((next_T <= T_STOP).to(torch.int8), dim=1).values > 0
            contrib = valid & ~stopped
        alpha_c = torch.where(contrib, alpha, torch.zeros_like(alpha))
        cp = torch.cumprod(1 - alpha_c, dim=1)
        T_excl = torch.cat([torch.ones_like(cp[:, :1]), cp[:, :-1]], dim=1)
        w = alpha_c * T_excl
        pix = w @ col
        T_final = cp[:, -1]

        if absgrad_buf is not None and alpha_c.requires_grad:
            def _hook(v_alpha, g=g, dx=dx.detach(), dy=dy.detach(), a=a.detach(), b=b.detach(),
                      c=c.detach(), alpha_raw=alpha_raw.detach(), contrib=contrib):
                live = contrib & (alpha_raw <= ALPHA_MAX)
                v_sigma = torch.where(live, -alpha_raw * v_alpha, torch.zeros_like(v_alpha))
                vx = (v_sigma * (a * dx + b * dy)).abs().sum(0)
                vy = (v_sigma * (b * dx + c * dy)).abs().sum(0)
                absgrad_buf.index_add_(0, g, torch.stack([vx, vy], dim=-1))
                return None
            alpha_c.register_hook(_hook)

        with torch.no_grad():
            idx = torch.arange(s, e, dtype=torch.int32)[None, :].expand_as(contrib)
            last = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=1).values
            last_ids[i0:i1, j0:j1] = last.reshape(i1 - i0, j1 - j0)
        out_r.append((i0, i1, j0, j1, pix))
        out_a.append((i0, i1, j0, j1, 1 - T_final))

    if out_r:
        # assemble with one differentiable scatter each
        rows = torch.cat([
            (torch.arange(i0, i1)[:, None] * width + torch.arange(j0, j1)[None, :]).reshape(-1)
            for (i0, i1, j0, j1, _) in out_r])
        render = render.reshape(-1, D).index_put((rows,), torch.cat([p for *_, p in out_r]))
        render = render.reshape(height, width, D)
        alphas = alphas.reshape(-1).index_put((rows,), torch.cat([p for *_, p in out_a]))
        alphas = alphas.reshape(height, width, 1)
    return render, alphas, last_ids


# --------------------------------------------------------------------------------------
# the boundary: same signature, same returns as the call at edge_gs.py:250-268
# --------------------------------------------------------------------------------------
def rasterization(
    means, quats, scales, opacities, colors, viewmats, Ks, width, height,
    near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=EPS2D, sh_degree=None,
    packed=False, tile_size=16, backgrounds=None, render_mode="RGB", sparse_grad=False,
    absgrad=False, rasterize_mode="classic", channel_chunk=32,
):
    """CPU oracle of ``gsplat.rasterization`` for the argument subset the reference uses
    (C cameras looped, packed=False, sh_degree=None, backgrounds=None, render_mode='RGB')."""
    assert sh_degree is None and backgrounds is None and render_mode == "RGB" and not packed
    assert rasterize_mode in ("classic", "antialiased")
    C = viewmats.shape[0]
    N = means.shape[0]
    tw = math.ceil(width / float(tile_size))
    th = math.ceil(height / float(tile_size))
    renders, alphas_l = [], []
    info_l: Dict[str, list] = {k: [] for k in (
        "radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss",
        "isect_ids", "flatten_ids", "isect_offsets", "last_ids")}
    tile_bits = int(math.floor(math.log2(tw * th))) + 1
    m_base = 0
    proj = [project(means, quats, scales, viewmats[cam], Ks[cam], width, height,
                    near_plane, far_plane, eps2d, radius_clip) for cam in range(C)]
    # [C,N,2] non-leaf tensor the compositing reads from, so that retain_grad()/.absgrad on
    # info["means2d"] behave as they do with gsplat (edge_gs.py:270-275,612)
    stacked_m2d = torch.stack([p[1] for p in proj])
    bufs = []
    for cam in range(C):
        radii, _, depths, conics, comp = proj[cam]
        m2d = stacked_m2d[cam]
        op = opacities * comp if rasterize_mode == "antialiased" else opacities
        tpg, ids, flat = isect_tiles(
            m2d.detach().numpy(), radii.numpy(), depths.detach().numpy(), tile_size, tw, th)
        offs = isect_offset_encode(ids, tw, th)
        col = colors if colors.dim() == 2 else colors[cam]
        buf = torch.zeros(N, 2, dtype=means.dtype) if absgrad else None
        bufs.append(buf)
        r, a, last = composite(m2d, conics, col, op, width, height, tile_size, offs, flat, buf)
        renders.append(r)
        alphas_l.append(a)
        info_l["radii"].append(radii)
        info_l["depths"].append(depths)
        info_l["conics"].append(conics)
        info_l["opacities"].append(op)
        info_l["tiles_per_gauss"].append(torch.from_numpy(tpg))
        info_l["isect_ids"].append(torch.from_numpy(ids | (cam << (32 + tile_bits))))
        info_l["flatten_ids"].append(torch.from_numpy(flat.astype(np.int32) + cam * N))
        info_l["isect_offsets"].append(torch.from_numpy(offs.astype(np.int32) + m_base))
        info_l["last_ids"].append(last + 0)
        m_base += int(flat.shape[0])

    if absgrad and stacked_m2d.requires_grad:
        def _set_absgrad(grad, t=stacked_m2d):
            t.absgrad = torch.stack(bufs).clone()
            return None
        stacked_m2d.register_hook(_set_absgrad)
    info = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": torch.stack(info_l["radii"]),
        "means2d": stacked_m2d,
        "depths": torch.stack(info_l["depths"]),
        "conics": torch.stack(info_l["conics"]),
        "opacities": torch.stack(info_l["opacities"]),
        "tile_width": tw, "tile_height": th,
        "tiles_per_gauss": torch.stack(info_l["tiles_per_gauss"]),
        "isect_ids": torch.cat(info_l["isect_ids"]),
        "flatten_ids": torch.cat(info_l["flatten_ids"]),
        "isect_offsets": torch.stack(info_l["isect_offsets"]),
        "last_ids": torch.stack(info_l["last_ids"]),
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
    }
    return torch.stack(renders), torch.stack(alphas_l), info


# --------------------------------------------------------------------------------------
# per-step glue around the call (edge_gs.py:278-324, losses.py, train_gaussians.py:83-106)
# --------------------------------------------------------------------------------------
def loss_weight_map(strategy: str, edge_mask: torch.Tensor,
                    bg_sel_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Every projection-loss strategy of edge_gs.py:288-324 is sum_p w_p |out_p - gt_p|.

    whole    : w = 1/HW                                        (edge_gs.py:290-296)
    weighted : w = weight_mask/HW, weight_mask from :177-193    (:316-319, losses.py:9-11)
    bg_edge_ratio : w = edge/#edge + sel/#sel, ``sel`` = the sampled mask built at :303-310
                    (drawn by the CALLER: it uses the CPU default RNG, so it is an input)."""
    hw = edge_mask.numel()
    if strategy == "whole":
        return torch.full(edge_mask.shape, 1.0 / hw, dtype=torch.float32)
    if strategy == "weighted":
        n_e = edge_mask.sum()
        n_b = (~edge_mask).sum()
        wm = torch.zeros(edge_mask.shape, dtype=torch.float32)
        wm[edge_mask] = (n_b / (n_e + n_b)).float()
        wm[~edge_mask] = (n_e / (n_e + n_b)).float()
        return wm / hw
    if strategy == "bg_edge_ratio":
        assert bg_sel_mask is not None
        w = edge_mask.float() / max(int(edge_mask.sum()), 1)
        w = w + bg_sel_mask.float() / max(int(bg_sel_mask.sum()), 1)
        return w
    raise ValueError(strategy)


def sample_bg_mask(edge_mask: torch.Tensor, ratio: float,
                   generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The sampled 'background' mask exactly as edge_gs.py:298-310 builds it, quirk included:
    ``torch.where(bg_mask)[0]`` is the ROW-index list (length #bg), a randperm of that LENGTH
    is taken, and the permutation values themselves (in [0,#bg)) are unravelled over H x W."""
    num_bg = int(ratio * edge_mask.sum())
    n = int((~edge_mask).sum())
    sel = torch.randperm(n, generator=generator)[:num_bg]
    H, W = edge_mask.shape
    sel = sel % (H * W)
    out = torch.zeros_like(edge_mask, dtype=torch.bool)
    out[sel // W, sel % W] = True
    return out


def edge_step_loss(render_ch0: torch.Tensor, gt: torch.Tensor, wmap: torch.Tensor) -> torch.Tensor:
    """clamp (edge_gs.py:279) + weighted L1 in weight-map form."""
    return (wmap * (torch.clamp(render_ch0, 0.0, 1.0) - gt).abs()).sum()


def adam_reference(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch 1.13 single-tensor Adam (the reference's pin, requirements.txt:221), out of place."""
    m = m * b1 + grads * (1 - b1)
    v = v * b2 + grads * grads * (1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return params - (lr / bc1) * m / denom, m, v


def filter_by_projection(means, edge_images, cameras, visib_thresh=0.1):
    """Restatement of edge_extraction/filtering.py:80-123 (numpy): x = K (R X + t), divide by the last
    row, np.round, bounds check, sum of the edge strengths at the hit pixels, mean over ALL views,
    strict > threshold.  Returns (inlier mask [N], mean visibility [N])."""
    import numpy as np
    means = np.asarray(means)
    n, v = means.shape[0], len(edge_images)
    vis = np.zeros((n, v))
    for i in range(v):
        K, R, t = cameras[i]["K"], cameras[i]["R"], cameras[i]["t"]
        h, w = cameras[i]["h"], cameras[i]["w"]
        x = (K @ (R @ means.reshape(-1, 3).T + t)).T
        uv = np.round((x / x[:, -1:])[:, :2]).astype(np.int32)
        ok = (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
        emap = np.asarray(edge_images[i])
        vis[ok, i] += emap[uv[ok, 1], uv[ok, 0]]
    mean_vis = vis.mean(axis=1)
    return mean_vis > visib_thresh, mean_vis
