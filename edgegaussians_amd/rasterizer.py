"""`rasterization(...)` -- the drop-in for the one call the reference makes into gsplat.

Mirrors, argument for argument, the operator the reference binds at
``edgegaussians/models/edge_gs.py:8`` and calls at ``edge_gs.py:250-268`` (gsplat 1.0.0's
``rasterization``), returning ``(render_colors[C,H,W,D], render_alphas[C,H,W,1], info)`` with
the same ``info`` keys.  What the caller reads back:

* ``info["means2d"]`` -- a non-leaf ``[C,N,2]`` tensor that requires grad (``retain_grad()``
  succeeds, edge_gs.py:270-271) and that carries ``.absgrad`` ``[C,N,2]`` after backward
  (edge_gs.py:612);
* ``info["radii"]`` -- int32 ``[C,N]`` (edge_gs.py:275).

Autograd layout is the same two-node chain as gsplat's (projection -> ``opacities *
compensations`` in torch -> compositing), so ``.absgrad`` is attached to the tensor the caller
holds.  All arithmetic runs in libedgegs.so (hand-written HIP, gfx950) through the C ABI of
include/edgegs.h with raw device pointers on the current HIP stream.  No CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import call, ptr, stream

TILE = 16


def _check(t: Tensor, shape, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (got {t.device}); edgegaussians_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")


class _Projection(torch.autograd.Function):
    """gsplat ``fully_fused_projection`` (packed=False) for C cameras, tile counting fused in."""

    @staticmethod
    def forward(ctx, means, quats, scales, opac_for_splat, viewmats, Ks, width, height, eps2d,
                near_plane, far_plane, radius_clip, antialiased):
        Cn, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        tw, th = math.ceil(width / TILE), math.ceil(height / TILE)
        means_c, quats_c, scales_c = means.contiguous(), quats.contiguous(), scales.contiguous()
        opac_c = opac_for_splat.contiguous()
        vm, Kc = viewmats.contiguous(), Ks.contiguous()
        splat = torch.empty(Cn, N, 8, device=dev)
        radii = torch.empty(Cn, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(Cn, N, 2, device=dev)
        depths = torch.empty(Cn, N, device=dev)
        conics = torch.empty(Cn, N, 3, device=dev)
        comps = torch.empty(Cn, N, device=dev)
        tpg = torch.empty(Cn, N, dtype=torch.int32, device=dev)
        counts = torch.zeros(Cn, tw * th, dtype=torch.int32, device=dev)
        flags = _lib.FLAG_ANTIALIASED if antialiased else 0
        for c in range(Cn):
            call("eg_project_fwd", ptr(means_c), ptr(quats_c), ptr(scales_c), ptr(opac_c), ptr(vm[c]), ptr(Kc[c]),
                 N, width, height, near_plane, far_plane, eps2d, radius_clip, flags,
                 ptr(splat[c]), ptr(radii[c]), ptr(means2d[c]), ptr(depths[c]), ptr(conics[c]), ptr(comps[c]),
                 ptr(tpg[c]), ptr(counts[c]), None, stream())
        ctx.save_for_backward(means_c, quats_c, scales_c, opac_c, vm, Kc, splat)
        ctx.cfg = (width, height, eps2d, flags)
        ctx.mark_non_differentiable(radii, tpg, counts, splat)
        return radii, means2d, depths, conics, comps, tpg, counts, splat

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, _a, _b, _c):
        means, quats, scales, opac, vm, Kc, splat = ctx.saved_tensors
        width, height, eps2d, flags = ctx.cfg
        Cn, N = vm.shape[0], means.shape[0]
        dev = means.device
        base = getattr(v_means2d, "_base", None) if v_means2d is not None else None
        if (base is not None and v_conics is not None and getattr(v_conics, "_base", None) is base
                and tuple(base.shape) == (Cn, N, 8) and base.is_contiguous() and base.dtype == torch.float32
                and v_means2d.storage_offset() == base.storage_offset()
                and v_conics.storage_offset() == base.storage_offset() + 4):
            g2d = base  # both are views of the compositing backward's packed record: no re-pack
        else:
            z2 = torch.zeros(Cn, N, 2, device=dev)
            z1 = torch.zeros(Cn, N, 1, device=dev)
            vm2d = v_means2d if v_means2d is not None else z2
            vcon = v_conics if v_conics is not None else torch.zeros(Cn, N, 3, device=dev)
            g2d = torch.cat([vm2d, z2, vcon, z1], dim=-1).contiguous()
        vcomp = (v_comps if v_comps is not None else torch.zeros(Cn, N, device=dev)).contiguous()
        vdep = v_depths.contiguous() if v_depths is not None else None
        v_means = torch.empty(N, 3, device=dev)
        v_quats = torch.empty(N, 4, device=dev)
        v_scales = torch.empty(N, 3, device=dev)
        tm, tq, ts = (v_means, v_quats, v_scales) if Cn == 1 else (torch.empty_like(v_means), torch.empty_like(v_quats),
                                                                  torch.empty_like(v_scales))
        for c in range(Cn):
            call("eg_project_bwd", ptr(means), ptr(quats), ptr(scales), ptr(opac), ptr(vm[c]), ptr(Kc[c]),
                 N, width, height, eps2d, flags, ptr(splat[c]), ptr(g2d[c]), ptr(vcomp[c]),
                 ptr(vdep[c]) if vdep is not None else None, ptr(tm), ptr(tq), ptr(ts), None, None, stream())
            if Cn > 1:
                if c == 0:
                    v_means.copy_(tm); v_quats.copy_(tq); v_scales.copy_(ts)
                else:
                    v_means += tm; v_quats += tq; v_scales += ts
        return (v_means, v_quats, v_scales) + (None,) * 10


class _Compositing(torch.autograd.Function):
    """gsplat ``rasterize_to_pixels`` (packed=False, backgrounds=None) for C cameras."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, width, height, offsets, flatten_ids, absgrad,
                unit_colors, item_offsets, totals, n_items, packed_splat):
        Cn, N = means2d.shape[0], means2d.shape[1]
        dev = means2d.device
        D = colors.shape[-1]
        # the projection kernel already wrote the packed record (x y a b c o*comp depth radius) the compositing
        # kernels read: the same floats as (means2d, conics, opacities), so no torch.cat re-pack per call
        splat = packed_splat
        colors_c = colors.contiguous()
        render = torch.empty(Cn, height, width, D, device=dev)
        alphas = torch.empty(Cn, height, width, 1, device=dev)
        last_ids = torch.empty(Cn, height, width, dtype=torch.int32, device=dev)
        # unit colours: the sliced forward leaves the per-pixel record {T_final, stop id, stop depth} that the
        # order-independent footprint backward reads (deterministic, no atomics, no zero-fill of the gradients)
        sliced = unit_colors and all(n > 0 for n in n_items)
        gtstop = torch.empty(Cn, height, width, 3, device=dev) if sliced else None
        for c in range(Cn):
            col = None if unit_colors else ptr(colors_c[c] if colors_c.dim() == 3 else colors_c)
            ws = None
            if unit_colors and n_items[c] > 0:  # slice-parallel forward needs its scratch
                ws = _lib.composite_workspace(n_items[c], offsets[c].shape[0] - 1, dev)
            call("eg_composite_fwd", ptr(splat[c]), col, D, ptr(offsets[c]), ptr(flatten_ids[c]), width, height,
                 ptr(render[c]), ptr(alphas[c]), ptr(last_ids[c]), None, None, 1.0, None, None,
                 ptr(item_offsets[c]) if ws is not None else None, ptr(totals[c]) if ws is not None else None,
                 n_items[c], ptr(ws), ptr(gtstop[c]) if sliced else None, -1, stream())
        ctx.save_for_backward(means2d, splat, colors_c, alphas, last_ids, *offsets, *flatten_ids, *item_offsets,
                              *totals)
        ctx.gtstop = gtstop
        ctx.cfg = (width, height, absgrad, unit_colors, Cn, tuple(n_items))
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_last):
        width, height, absgrad, unit_colors, Cn, n_items = ctx.cfg
        saved = ctx.saved_tensors
        means2d, splat, colors, alphas, last_ids = saved[:5]
        offsets, flatten_ids = saved[5:5 + Cn], saved[5 + Cn:5 + 2 * Cn]
        item_offsets, totals = saved[5 + 2 * Cn:5 + 3 * Cn], saved[5 + 3 * Cn:5 + 4 * Cn]
        N, D = means2d.shape[1], colors.shape[-1]
        dev = means2d.device
        need_vcol = ctx.needs_input_grad[2]
        v_colors = None
        if unit_colors and not need_vcol and ctx.gtstop is not None:
            # footprint backward: every Gaussian sums over its own footprint in the {v * T_final, stop} record
            rec = ctx.gtstop.clone()
            rec[..., 0] *= (v_render.sum(-1) + v_alphas[..., 0])
            g2d = torch.empty(Cn, N, 8, device=dev)
            for c in range(Cn):
                call("eg_composite_bwd_footprint", ptr(splat[c]), N, width, height, ptr(rec[c]), ptr(g2d[c]), stream())
            if absgrad:
                means2d.absgrad = g2d[..., 2:4].contiguous()
            # views of the one [C,N,8] record: the projection backward recognises them and skips the re-pack
            return (g2d[..., 0:2], g2d[..., 4:7], None, g2d[..., 7]) + (None,) * 10
        g2d = torch.zeros(Cn, N, 8, device=dev)
        v_render = v_render.contiguous()
        v_alphas = v_alphas.contiguous()
        if unit_colors and not need_vcol:
            vpix = (v_render.sum(-1) + v_alphas[..., 0]).contiguous()
            for c in range(Cn):
                call("eg_composite_bwd", ptr(splat[c]), ptr(offsets[c]), ptr(flatten_ids[c]), width, height,
                     ptr(alphas[c]), ptr(last_ids[c]), ptr(vpix[c]), ptr(g2d[c]),
                     ptr(item_offsets[c]), ptr(totals[c]), n_items[c], stream())
        else:
            per_cam = colors.dim() == 3
            v_colors = torch.zeros(Cn, N, D, device=dev) if need_vcol else None
            for c in range(Cn):
                call("eg_composite_bwd_colors", ptr(splat[c]), ptr(colors[c] if per_cam else colors), D,
                     ptr(offsets[c]), ptr(flatten_ids[c]), width, height, ptr(alphas[c]), ptr(last_ids[c]),
                     ptr(v_render[c]), ptr(v_alphas[c]), ptr(g2d[c]),
                     ptr(v_colors[c]) if v_colors is not None else None, stream())
            if v_colors is not None and not per_cam:
                v_colors = v_colors.sum(0)
        if absgrad:
            means2d.absgrad = g2d[..., 2:4].contiguous()
        return (g2d[..., 0:2].contiguous(), g2d[..., 4:7].contiguous(), v_colors, g2d[..., 7].contiguous(),
                None, None, None, None, None, None, None, None, None, None)


def isect_tiles_and_sort(means2d: Tensor, radii: Tensor, depths: Tensor, counts: Tensor, width: int,
                         height: int, want_isect_ids: bool = True, max_tile_hint: Optional[int] = None,
                         extra_flag: Optional[Tensor] = None
                         ) -> Tuple[Tensor, Tensor, Optional[Tensor], int, Tensor, Tensor, int]:
    """Per camera: offsets[T+1] (scan of `counts`), keys -> sorted flatten_ids (+ int64 isect ids).

    `counts` holds the per-tile counts on entry and is returned to zero.  One host sync (the read
    of M), exactly where gsplat has one (the cumsum total before allocating the isect arrays)."""
    dev = means2d.device
    N = means2d.shape[0]
    T = counts.shape[0]
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    item_offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    total = torch.zeros(4, dtype=torch.int32, device=dev)  # total[1] (overflow) is sticky: start from zero
    call("eg_tile_offsets", ptr(counts), T, 1 << 40, ptr(offsets), ptr(item_offsets), ptr(total), stream())
    if extra_flag is not None:  # a device scalar the caller wants back with this (only) read-back
        vals = torch.cat([total, extra_flag.to(torch.int32).reshape(1)]).tolist()
        isect_tiles_and_sort.last_extra = int(vals[4])
        M, _ovf, n_items, nmax = (int(v) for v in vals[:4])
    else:
        M, _ovf, n_items, nmax = (int(v) for v in total.tolist())
    keys = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    flat = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    ids = torch.empty(max(M, 1), dtype=torch.int64, device=dev) if want_isect_ids else None
    call("eg_tile_emit", ptr(means2d), ptr(radii), ptr(depths), None, 0, N, width, height, ptr(offsets),
         ptr(counts), M, ptr(keys), None, stream())
    call("eg_sort_pairs", ptr(keys), ptr(offsets), T, M, ptr(flat), ptr(ids) if ids is not None else None,
         nmax if max_tile_hint is None else max_tile_hint, stream())  # the host knows the exact maximum here
    return offsets, flat[:M], (ids[:M] if ids is not None else None), M, item_offsets, total, n_items


def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
    viewmats: Tensor, Ks: Tensor, width: int, height: int,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
    sh_degree: Optional[int] = None, packed: bool = True, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic", channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    """Same names, argument meaning and defaults as gsplat 1.0.0 ``rasterization``; supports the
    subset the reference reaches (edge_gs.py:250-268): packed=False, sh_degree=None,
    backgrounds=None, render_mode='RGB', tile_size=16, colours of 1 or 3 channels."""
    N = means.shape[0]
    Cn = viewmats.shape[0]
    _check(means, (N, 3), "means")
    _check(quats, (N, 4), "quats")
    _check(scales, (N, 3), "scales")
    _check(opacities, (N,), "opacities")
    _check(viewmats, (Cn, 4, 4), "viewmats")
    _check(Ks, (Cn, 3, 3), "Ks")
    if render_mode != "RGB":
        raise NotImplementedError("render_mode other than 'RGB' is outside the reference's path")
    if sh_degree is not None or backgrounds is not None:
        raise NotImplementedError("sh_degree / backgrounds are outside the reference's path")
    if packed or sparse_grad:
        raise NotImplementedError("packed / sparse_grad are outside the reference's path (edge_gs.py:261,265)")
    if tile_size != TILE:
        raise NotImplementedError("tile_size must be 16 (edge_gs.py:232)")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if colors.dim() == 2:
        _check(colors, (N, colors.shape[-1]), "colors")
    else:
        _check(colors, (Cn, N, colors.shape[-1]), "colors")
    D = colors.shape[-1]
    if D not in (1, 3):
        raise NotImplementedError("colors must have 1 or 3 channels")
    width, height = int(width), int(height)
    antialiased = rasterize_mode == "antialiased"

    radii, means2d, depths, conics, comps, tpg, counts, _splat = _Projection.apply(
        means, quats, scales, opacities.detach(), viewmats, Ks, width, height, float(eps2d),
        float(near_plane), float(far_plane), float(radius_clip), antialiased)
    opac = opacities[None, :].expand(Cn, N)
    if antialiased:
        opac = opac * comps

    tw, th = math.ceil(width / TILE), math.ceil(height / TILE)
    tile_bits = int(math.floor(math.log2(tw * th))) + 1
    offs_l, flat_l, ids_l, item_l, tot_l, nit_l = [], [], [], [], [], []
    m_base = 0
    info_offsets = []
    # the reference always passes torch.ones(N,3) without grad (edge_gs.py:247, a fresh tensor every step): decide
    # "unit colours" on the device and read the verdict back with M (the one host sync gsplat also has)
    unit_flag = None if colors.requires_grad else (colors == 1).all()
    with torch.no_grad():
        for c in range(Cn):
            offsets, flat, ids, M, item_offsets, total, n_items = isect_tiles_and_sort(
                means2d[c], radii[c], depths[c], counts[c], width, height, extra_flag=unit_flag if c == 0 else None)
            item_l.append(item_offsets)
            tot_l.append(total)
            nit_l.append(n_items)
            offs_l.append(offsets)
            flat_l.append(flat if M > 0 else torch.zeros(1, dtype=torch.int32, device=means.device))
            ids_l.append((ids | (c << (32 + tile_bits)), flat + c * N, M))
            info_offsets.append((offsets[:-1] + m_base).reshape(th, tw))
            m_base += M

    # ... and take the order-independent unit-colour kernels when that is what we were given
    unit = unit_flag is not None and bool(isect_tiles_and_sort.last_extra)
    render, alphas, last_ids = _Compositing.apply(
        means2d, conics, colors, opac.contiguous(), width, height, tuple(offs_l), tuple(flat_l), bool(absgrad),
        unit, tuple(item_l), tuple(tot_l), tuple(nit_l), _splat)

    info = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics, "opacities": opac,
        "tile_width": tw, "tile_height": th, "tiles_per_gauss": tpg,
        "isect_ids": torch.cat([t[0][:t[2]] for t in ids_l]),
        "flatten_ids": torch.cat([t[1][:t[2]] for t in ids_l]),
        "isect_offsets": torch.stack(info_offsets),
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": Cn,
        "last_ids": last_ids,
    }
    return render, alphas, info
