"""Compositing kernels fed with the ORACLE's floats and the ORACLE's bins (C-ABI level).

north_star: "tile/pixel indexing bit-exact".  Whether a pixel's last contributor, its alpha and the
per-Gaussian 2-D gradients agree can only be asked on identical inputs: these tests hand the C oracle's
projected means2d / conics / opacities and its sorted tile lists to `eg_composite_fwd` (the classic tile
kernel AND the slice -> combine -> re-walk path), `eg_composite_bwd` (operator path) and
`eg_composite_bwd_footprint` (fused path), and compare

  * last_ids  -- EXACT on every pixel outside the quantified borderline set (tests/util.py),
  * alphas    -- 1e-4 on the same pixels,
  * g2d       -- 1e-4 on EVERY Gaussian, the borderline pixels having zero loss weight on both sides.
"""
import math

import numpy as np
import pytest
import torch

from tests.util import (REL_ALPHA, REL_T, assert_close, borderline_pixel_mask, masked_weights, oracle_forward, record,
                        rel_err, to_np)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from edgegaussians_amd import _lib
    _lib.load()
    from edgegaussians_amd import synth
    from oracle import c_oracle as CO
    return _lib, synth, CO


def _scene(synth, kind):
    if kind == "spread":  # mixed opacities, a few hundred Gaussians per tile
        return synth.make_scene(3000, 1, 200, 136, seed=0, spread_opacity=True, scale=0.02, anisotropy=5.0)
    if kind == "stops":   # opaque and heavily overlapping: most pixels end on the transmittance stop
        sc = synth.make_scene(4000, 1, 128, 96, seed=6, spread_opacity=False, scale=0.03, anisotropy=2.0)
        sc.logit_opacities[:] = torch.logit(torch.tensor(0.97))
        return sc
    if kind == "deep":    # > 128 Gaussians in most tiles AND stops: every slice-parallel code path
        sc = synth.make_scene(12000, 1, 96, 80, seed=8, spread_opacity=True, scale=0.03, anisotropy=3.0)
        return sc
    raise ValueError(kind)


def _device_inputs(fw):
    """splat [N,8] (x y a b c o depth radius-bits), offsets [T+1], flatten_ids on the device, from oracle floats."""
    N = fw["means2d"].shape[0]
    sp = np.zeros((N, 8), np.float32)
    sp[:, 0:2] = fw["means2d"]
    sp[:, 2:5] = fw["conics"]
    sp[:, 5] = fw["opacities"]
    sp[:, 6] = fw["depths"]
    sp[:, 7] = fw["radii"].view(np.float32)
    offs = np.concatenate([fw["isect_offsets"].reshape(-1), np.array([fw["M"]], np.int32)]).astype(np.int32)
    flat = fw["flatten_ids"] if fw["M"] > 0 else np.zeros(1, np.int32)
    return (torch.from_numpy(sp).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(np.ascontiguousarray(flat)).cuda())


def _items(offsets_dev, T):
    """item table of the slice-parallel kernels from the oracle's offsets (eg_tile_offsets on the counts)."""
    from edgegaussians_amd._lib import call, ptr, stream
    counts = (offsets_dev[1:] - offsets_dev[:-1]).to(torch.int32).contiguous()
    offs2 = torch.empty(T + 1, dtype=torch.int32, device="cuda")
    item_offsets = torch.empty(T + 1, dtype=torch.int32, device="cuda")
    total = torch.zeros(4, dtype=torch.int32, device="cuda")
    call("eg_tile_offsets", ptr(counts), T, 1 << 40, ptr(offs2), ptr(item_offsets), ptr(total), stream())
    assert torch.equal(offs2, offsets_dev)
    return item_offsets, total, int(total[2].item())


def _composite_fwd(_lib, splat, offs, flat, W, H, sliced, gt=None, wmap=None):
    from edgegaussians_amd._lib import call, ptr, stream
    T = math.ceil(W / 16) * math.ceil(H / 16)
    render = torch.zeros(H, W, 1, device="cuda")
    alphas = torch.zeros(H, W, device="cuda")
    last = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
    vpix = torch.zeros(H, W, device="cuda") if wmap is not None else None
    loss = torch.zeros(1, device="cuda") if wmap is not None else None
    gtstop = torch.zeros(H, W, 3, device="cuda") if (wmap is not None and sliced) else None
    item_offsets = total = ws = None
    n_items = 0
    if sliced:
        item_offsets, total, n_items = _items(offs, T)
        ws = _lib.composite_workspace(max(n_items, 1), T, "cuda")
    call("eg_composite_fwd", ptr(splat), None, 1, ptr(offs), ptr(flat), W, H, ptr(render), ptr(alphas), ptr(last),
         ptr(gt), ptr(wmap), 1.0, ptr(vpix), ptr(loss), ptr(item_offsets), ptr(total), max(n_items, 1) if sliced else 0,
         ptr(ws), ptr(gtstop), -1, stream())
    torch.cuda.synchronize()
    return dict(render=render[..., 0], alphas=alphas, last=last, vpix=vpix, loss=loss, gtstop=gtstop,
                item_offsets=item_offsets, total=total, n_items=n_items)


@pytest.mark.parametrize("kind", ["spread", "stops", "deep"])
@pytest.mark.parametrize("sliced", [False, True])
def test_composite_forward_pixel_indexing_exact_on_oracle_floats(env, kind, sliced):
    _lib, synth, CO = env
    sc = _scene(synth, kind)
    W, H = sc.width, sc.height
    fw = oracle_forward(sc, 0)
    assert fw["M"] > 0
    border = borderline_pixel_mask(fw)
    frac = float(border.float().mean())
    assert frac < 0.03, f"borderline set too large to be a useful exclusion: {frac}"
    splat, offs, flat = _device_inputs(fw)
    out = _composite_fwd(_lib, splat, offs, flat, W, H, sliced)
    ok = ~border
    last_g, last_o = out["last"].cpu(), torch.from_numpy(fw["last_ids"])
    a_g, a_o = out["alphas"].cpu(), torch.from_numpy(fw["alphas"])
    n_bad_idx = int((last_g[ok] != last_o[ok]).sum())
    e_alpha = rel_err(a_g[ok], a_o[ok])
    # how the two agree on the excluded pixels, for the record (not asserted)
    n_border_diff = int((last_g[border] != last_o[border]).sum())
    record("composite_fwd_on_oracle_floats", scene=kind, path="sliced" if sliced else "tile", pixels=int(ok.numel()),
           borderline_pixels=int(border.sum()), last_id_mismatches_outside_borderline=n_bad_idx,
           last_id_mismatches_inside_borderline=n_border_diff, alpha_max_rel_err=e_alpha,
           stopped_frac=float((a_o > 1 - 1.1e-4).float().mean()), largest_tile=int(np.diff(to_np(offs)).max()))
    assert n_bad_idx == 0, f"{n_bad_idx} pixels outside the borderline set disagree on the last contributor"
    assert_close(a_g[ok], a_o[ok], rtol=1e-4, name="alphas")
    assert_close(out["render"].cpu()[ok], torch.from_numpy(fw["render"][..., 0])[ok], rtol=1e-4, name="render")
    if kind == "stops":
        assert float((a_o > 1 - 1.1e-4).float().mean()) > 0.03, "scene must saturate a share of the pixels"
    if kind == "deep":
        assert int(np.diff(to_np(offs)).max()) > 256, "scene must put several slices into one tile"


@pytest.mark.parametrize("kind", ["spread", "stops", "deep"])
def test_composite_backward_on_oracle_floats(env, kind):
    """2-D gradients (v_means2d, |v_means2d|, v_conics, v_opacity) of both backward kernels against the C
    oracle's sequential back-to-front walk, same floats, same bins, borderline pixels zero-weighted."""
    _lib, synth, CO = env
    from edgegaussians_amd._lib import call, ptr, stream
    sc = _scene(synth, kind)
    W, H, N = sc.width, sc.height, sc.means.shape[0]
    fw = oracle_forward(sc, 0)
    gt = sc.gt[0]
    border = borderline_pixel_mask(fw, gt)
    w = masked_weights(synth.weight_map("weighted", gt), border)
    # oracle: upstream gradient of the clamp + weighted L1 from ITS render, then its backward walk
    ro = torch.from_numpy(fw["render"][..., 0])
    d = torch.clamp(ro, 0, 1) - gt
    v_render = (w * torch.sign(d)).numpy()[..., None].astype(np.float32)
    want = CO.backward(fw, v_render)
    loss_o = float((w.double() * d.abs().double()).sum())
    ref = np.concatenate([want["means2d"], want["absgrad"], want["conics"], want["opacities_eff"][:, None]], axis=1)
    splat, offs, flat = _device_inputs(fw)
    gt_d, w_d = gt.cuda().contiguous(), w.cuda().contiguous()
    # (i) fused path: sliced forward with the loss epilogue -> gtstop -> footprint backward
    out = _composite_fwd(_lib, splat, offs, flat, W, H, True, gt_d, w_d)
    assert abs(float(out["loss"]) - loss_o) <= 1e-4 * abs(loss_o)
    g2d_f = torch.full((N, 8), float("nan"), device="cuda")
    call("eg_composite_bwd_footprint", ptr(splat), N, W, H, ptr(out["gtstop"]), ptr(g2d_f), stream())
    # (ii) operator path: item-parallel backward from (alphas, last_ids, vpix)
    g2d_i = torch.zeros(N, 8, device="cuda")
    call("eg_composite_bwd", ptr(splat), ptr(offs), ptr(flat), W, H, ptr(out["alphas"]), ptr(out["last"]),
         ptr(out["vpix"]), ptr(g2d_i), ptr(out["item_offsets"]), ptr(out["total"]), max(out["n_items"], 1), stream())
    # (iii) operator path, one workgroup per tile
    g2d_t = torch.zeros(N, 8, device="cuda")
    call("eg_composite_bwd", ptr(splat), ptr(offs), ptr(flat), W, H, ptr(out["alphas"]), ptr(out["last"]),
         ptr(out["vpix"]), ptr(g2d_t), None, None, 0, stream())
    torch.cuda.synchronize()
    vis = torch.from_numpy(fw["radii"] > 0)
    names = ("v_means2d", "v_means2d_abs", "v_conics", "v_opacity")
    cols = ((0, 2), (2, 4), (4, 7), (7, 8))
    errs = {}
    for tag, g in (("footprint", g2d_f), ("item", g2d_i), ("tile", g2d_t)):
        g = g.cpu()
        for name, (c0, c1) in zip(names, cols):
            a, b = g[vis][:, c0:c1], torch.from_numpy(ref)[vis][:, c0:c1]
            errs[f"{tag}:{name}"] = rel_err(a, b)
            assert_close(a, b, rtol=1e-4, name=f"{tag} {name}")
    record("composite_bwd_on_oracle_floats", scene=kind, borderline_pixels=int(border.sum()), pixels=int(border.numel()),
           max_rel_err=errs, rel_alpha=REL_ALPHA, rel_T=REL_T)
