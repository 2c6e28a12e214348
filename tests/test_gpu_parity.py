"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (BASELINE.json north_star): floats 1e-4 relative; tile / pixel indexing bit-exact
(on identical float inputs -- the binning tests here and the compositing tests in
test_gpu_oracle_floats.py feed the ORACLE's floats to the HIP kernels).

Single-step comparisons assert 1e-4 on EVERY element: the float-borderline branches of the path are
handled by a quantified exclusion (tests/util.py) -- Gaussians whose integer decisions hinge on rounding
are taken out of the scene, pixels within a stated margin of a threshold get zero loss weight on both
sides.  Only the multi-step TRAJECTORY tests (protocol checks: step counts, alternation, absgrad
accumulation) keep a looser, documented tolerance, because after the first Adam step the two
implementations no longer hold bit-identical parameters.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.util import (assert_close, borderline_pixel_mask, check_fused_step_vs_c_oracle, clean_scene, masked_weights,
                        oracle_forward, oracle_raw_grads, record, rel_err, strict_inputs, to_np)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from edgegaussians_amd import _lib
    _lib.load()  # raises if the .so or the GPU is missing: no fallback
    from edgegaussians_amd import synth
    from oracle import ref_torch as O
    return _lib, synth, O


def _scene(synth, n=3000, w=200, h=136, seed=0, scale=0.02, views=2, aniso=5.0):
    return synth.make_scene(n, views, w, h, seed=seed, spread_opacity=True, scale=scale, anisotropy=aniso)


def _dev(*ts):
    return [t.cuda().contiguous() for t in ts]


# ------------------------------------------------------------------ G1
def test_projection_forward(env):
    _lib, synth, O = env
    from edgegaussians_amd._lib import call, ptr, stream
    sc = _scene(synth)
    # put some Gaussians behind the camera and some far off-screen to exercise the culls
    R, t = sc.viewmats[0, :3, :3], sc.viewmats[0, :3, 3]
    cam_c = -R.T @ t
    sc.means[:30] = cam_c - 2.0 * R[2] + 0.1 * sc.means[:30]
    sc.means[30:50] = cam_c + 2.0 * R[2] + 40.0 * R[0]
    scales = torch.exp(sc.log_scales)
    op = torch.sigmoid(sc.logit_opacities).squeeze(-1)
    N, W, H = sc.means.shape[0], sc.width, sc.height
    radii_o, m2d_o, dep_o, con_o, comp_o = O.project(sc.means, sc.quats, scales, sc.viewmats[0], sc.Ks[0], W, H)
    means, quats, scl, opd, vm, K = _dev(sc.means, sc.quats, scales, op, sc.viewmats[0], sc.Ks[0])
    splat = torch.empty(N, 8, device="cuda")
    radii = torch.empty(N, dtype=torch.int32, device="cuda")
    m2d = torch.empty(N, 2, device="cuda")
    dep = torch.empty(N, device="cuda")
    con = torch.empty(N, 3, device="cuda")
    comp = torch.empty(N, device="cuda")
    call("eg_project_fwd", ptr(means), ptr(quats), ptr(scl), ptr(opd), ptr(vm), ptr(K), N, W, H, 0.01, 1e10, 0.3,
         0.0, _lib.FLAG_ANTIALIASED, ptr(splat), ptr(radii), ptr(m2d), ptr(dep), ptr(con), ptr(comp), None, None,
         None, stream())
    torch.cuda.synchronize()
    r = to_np(radii)
    ro = to_np(radii_o)
    # STRICT form (round 4): the scene is NOT cleaned here -- instead the Gaussians whose integer decisions lie within
    # 2e-5 of a float threshold are identified (oracle/eg_oracle.c: ego_project_borderline, double precision on the
    # oracle's own fp32 values) and the two sets are held to different standards:
    #   outside the set: cull decisions and radii IDENTICAL for every Gaussian;
    #   inside the set:  the outcome is BOUNDED -- the radius differs by at most 1, and a cull decision may differ only
    #                    where the other side's radius is at most 1 ... or the Gaussian sits on the near plane / screen edge
    from oracle import c_oracle as CO
    border = CO.project_borderline(sc.means.numpy(), sc.quats.numpy(), scales.numpy(), sc.viewmats[0].numpy(),
                                   sc.Ks[0].numpy(), W, H) > 0
    assert border.mean() <= 0.02, border.mean()
    assert np.array_equal(r[~border], ro[~border]), f"{int((r[~border] != ro[~border]).sum())} radii differ outside the borderline set"
    both_b = border & (r > 0) & (ro > 0)
    assert np.abs(r[both_b] - ro[both_b]).max(initial=0) <= 1
    record("projection_forward_strict", gaussians=int(N), borderline=int(border.sum()),
           radius_mismatches_inside=int((r[border] != ro[border]).sum()), cull_flips_inside=int(((r > 0) != (ro > 0))[border].sum()))
    both = (r > 0) & (ro > 0)
    assert both.sum() > 1000 and (ro == 0).sum() >= 50
    sel = torch.from_numpy(both)
    assert_close(m2d.cpu()[sel], m2d_o[sel], name="means2d")
    assert_close(dep.cpu()[sel], dep_o[sel], name="depths")
    assert_close(con.cpu()[sel], con_o[sel], name="conics")
    assert_close(comp.cpu()[sel], comp_o[sel], name="compensations")
    # packed record agrees with the separate outputs
    sp = splat.cpu()
    assert torch.equal(sp[:, 0:2], m2d.cpu()) and torch.equal(sp[:, 2:5], con.cpu())
    assert torch.equal(sp[:, 7].view(torch.int32), radii.cpu())
    assert_close(sp[:, 5][sel], (op * comp_o)[sel], name="opacity*comp")


# ------------------------------------------------------------------ G2-G6 (bit exact on oracle floats)
def _bin_on_device(means2d, radii, depths, W, H, max_tile_hint=None):
    from edgegaussians_amd._lib import call, ptr, stream
    from edgegaussians_amd.rasterizer import isect_tiles_and_sort
    N = means2d.shape[0]
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    m, r, d = _dev(means2d, radii, depths)
    tpg = torch.empty(N, dtype=torch.int32, device="cuda")
    counts = torch.zeros(tw * th, dtype=torch.int32, device="cuda")
    call("eg_tile_count", ptr(m), ptr(r), N, W, H, ptr(tpg), ptr(counts), stream())
    counts_copy = counts.clone()
    offsets, flat, ids, M = isect_tiles_and_sort(m, r, d, counts, W, H, max_tile_hint=max_tile_hint)[:4]
    torch.cuda.synchronize()
    assert int(counts.abs().sum()) == 0, "emit must return the tile counters to zero"
    return to_np(tpg), to_np(ids), to_np(flat), to_np(offsets), M, to_np(counts_copy)


@pytest.mark.parametrize("case", ["scene", "ties", "huge_tile", "giant_tile", "empty", "mid_tile_stale_hint",
                                  "huge_tile_stale_hint"])
def test_binning_bit_exact(env, case):
    _lib, synth, O = env
    W, H = 200, 136
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    g = torch.Generator().manual_seed(3)
    if case == "scene":
        sc = _scene(synth, n=4000)
        radii, m2d, dep, _, _ = O.project(sc.means, sc.quats, torch.exp(sc.log_scales), sc.viewmats[1], sc.Ks[1], W, H)
    elif case == "ties":  # many equal depths: order must fall back to the Gaussian index
        n = 3000
        m2d = torch.rand(n, 2, generator=g) * torch.tensor([W, H])
        radii = torch.randint(1, 30, (n,), generator=g, dtype=torch.int32)
        dep = torch.randint(1, 5, (n,), generator=g).float()
    elif case in ("huge_tile", "giant_tile", "mid_tile_stale_hint", "huge_tile_stale_hint"):
        # huge: 8192 < n <= 16384 in one tile -> in-LDS bitonic network; giant: > 16384 -> hybrid
        # global/LDS network.  (scene: bucket+rank; ties: bucket overflow -> bitonic fallback)
        # *_stale_hint: the caller's tile-population hint (1) is far too small, so the large-tile launch is
        # skipped and the small variant must sort 2048 < n <= 4096 (in-LDS network) and n > 4096 (hybrid) itself
        n = {"huge_tile": 11000, "giant_tile": 21000, "mid_tile_stale_hint": 3400, "huge_tile_stale_hint": 11000}[case]
        m2d = torch.rand(n, 2, generator=g) * 10 + torch.tensor([40.0, 40.0])
        radii = torch.randint(1, 4, (n,), generator=g, dtype=torch.int32)
        dep = torch.rand(n, generator=g) * 5 + 0.5
        dep[::7] = dep[0]
    else:
        n = 64
        m2d = torch.zeros(n, 2)
        radii = torch.zeros(n, dtype=torch.int32)
        dep = torch.zeros(n)
    tpg_o, ids_o, flat_o = O.isect_tiles(to_np(m2d), to_np(radii), to_np(dep), 16, tw, th)
    offs_o = O.isect_offset_encode(ids_o, tw, th).reshape(-1)
    tpg, ids, flat, offsets, M, counts = _bin_on_device(m2d, radii, dep, W, H, 1 if case.endswith("stale_hint") else None)
    assert M == len(ids_o)
    if case == "mid_tile_stale_hint":
        assert 2048 < counts.max() <= 4096
    if case == "huge_tile":
        assert 8192 < counts.max() <= 16384
    if case == "giant_tile":
        assert counts.max() > 16384
    assert np.array_equal(tpg, tpg_o)
    assert np.array_equal(offsets[:-1], offs_o) and offsets[-1] == M
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(flat, flat_o)


# ------------------------------------------------------------------ full boundary call
def _run_pair(env, sc, view=0, colors=None, absgrad=True, mode="antialiased", loss_fn=None):
    _lib, synth, O = env
    from edgegaussians_amd import rasterization
    N = sc.means.shape[0]
    outs = []
    for dev in ("cpu", "cuda"):
        means = sc.means.clone().to(dev).requires_grad_(True)
        ls = sc.log_scales.clone().to(dev).requires_grad_(True)
        q = sc.quats.clone().to(dev).requires_grad_(True)
        lo = sc.logit_opacities.clone().to(dev).requires_grad_(True)
        col = (torch.ones(N, 3) if colors is None else colors.clone()).to(dev)
        if colors is not None:
            col.requires_grad_(True)
        fn = O.rasterization if dev == "cpu" else rasterization
        render, alpha, info = fn(
            means=means, quats=q, scales=torch.exp(ls), opacities=torch.sigmoid(lo).squeeze(-1), colors=col,
            viewmats=sc.viewmats[view:view + 1].to(dev), Ks=sc.Ks[view:view + 1].to(dev), width=sc.width,
            height=sc.height, tile_size=16, packed=False, near_plane=0.01, far_plane=1e10, render_mode="RGB",
            sparse_grad=False, absgrad=absgrad, rasterize_mode=mode)
        assert info["means2d"].requires_grad and not info["means2d"].is_leaf
        info["means2d"].retain_grad()
        loss = loss_fn(render, alpha, dev)
        loss.backward()
        outs.append(dict(render=render, alpha=alpha, info=info, means=means, ls=ls, q=q, lo=lo, col=col,
                         loss=loss))
    return outs


def _l1_loss(sc, w, view):
    def fn(render, alpha, dev):
        rgb = torch.clamp(render[0, ..., :3], 0.0, 1.0)  # edge_gs.py:278-279
        return (w.to(dev) * (rgb[:, :, 0] - sc.gt[view].to(dev)).abs()).sum()  # train_gaussians.py:84-94
    return fn


@pytest.mark.parametrize("unit", [True, False])
def test_rasterization_three_cameras_one_native_call_per_stage(env, unit, monkeypatch):
    """Round 6: gsplat.rasterization with viewmats [3, 4, 4] through the GENERAL operator path -- ONE
    native call per stage for the three cameras (csrc/cams.hip: projection, tile scans, emission + sort, compositing, the
    footprint backward, the projection backward summed over the cameras) and ONE host read-back -- against the dense PyTorch
    oracle (autograd backward) on the same three cameras: images per camera, gradients = the sum over the cameras.
    unit: colours all ones (the order-independent unit-colour kernels) / random per-Gaussian colours with a gradient."""
    _lib, synth, O = env
    from edgegaussians_amd import rasterization
    from edgegaussians_amd import rasterizer as R
    sc0 = _scene(synth, n=2500, views=5)
    cams = [0, 2, 3]
    sc, _removed = clean_scene(sc0, cams)
    N = sc.means.shape[0]
    ws = []
    for v in cams:
        fw = oracle_forward(sc, v)
        ws.append(masked_weights(synth.weight_map("weighted", sc.gt[v]), borderline_pixel_mask(fw, sc.gt[v])))
    g = torch.Generator().manual_seed(9)
    colors0 = torch.ones(N, 3) if unit else 0.2 + 0.8 * torch.rand(N, 3, generator=g)
    seen = []
    real_call = R.call
    monkeypatch.setattr(R, "call", lambda name, *a: (seen.append(name), real_call(name, *a))[1])
    outs = []
    for dev in ("cpu", "cuda"):
        p = [t.clone().to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
        col = colors0.clone().to(dev)
        if not unit:
            col.requires_grad_(True)
        fn = O.rasterization if dev == "cpu" else rasterization
        render, alpha, info = fn(means=p[0], quats=p[1], scales=torch.exp(p[2]), opacities=torch.sigmoid(p[3]).squeeze(-1), colors=col,
                                 viewmats=sc.viewmats[cams].to(dev), Ks=sc.Ks[cams].to(dev), width=sc.width, height=sc.height,
                                 tile_size=16, packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        loss = sum((ws[i].to(dev) * (torch.clamp(render[i, ..., 0], 0, 1) - sc.gt[v].to(dev)).abs()).sum() for i, v in enumerate(cams))
        loss.backward()
        outs.append(dict(render=render, alpha=alpha, info=info, p=p, col=col, loss=loss))
    cpu, gpu = outs
    assert gpu["render"].shape == (3, sc.height, sc.width, 3) and gpu["info"]["n_cameras"] == 3
    assert abs(float(gpu["loss"]) - float(cpu["loss"])) <= 1e-4 * abs(float(cpu["loss"]))
    for i in range(3):
        ok = ws[i] != 0
        assert_close(gpu["alpha"][i, ..., 0].detach().cpu()[ok], cpu["alpha"][i, ..., 0].detach()[ok], rtol=1e-4, name=f"alpha cam {i}")
        assert_close(gpu["render"][i].detach().cpu()[ok], cpu["render"][i].detach()[ok], rtol=1e-4, name=f"render cam {i}")
    assert np.array_equal(to_np(gpu["info"]["radii"]), to_np(cpu["info"]["radii"]))
    assert np.array_equal(to_np(gpu["info"]["isect_offsets"]), to_np(cpu["info"]["isect_offsets"]))
    for name, a, b in zip(("means", "quats", "scales", "opacities"), gpu["p"], cpu["p"]):
        assert_close(a.grad.cpu(), b.grad, rtol=1e-4, name=f"3 cameras grad {name}")
    assert_close(gpu["info"]["means2d"].absgrad.cpu(), cpu["info"]["means2d"].absgrad, rtol=1e-4, name="absgrad [3, N, 2]")
    if not unit:
        assert_close(gpu["col"].grad.cpu(), cpu["col"].grad, rtol=1e-4, name="3 cameras grad colors")
    # one native call per stage for the three cameras
    for stage in ("eg_project_fwd_cams", "eg_tile_offsets_cams", "eg_tile_emit_sort_cams", "eg_composite_fwd_cams", "eg_project_bwd_cams"):
        assert seen.count(stage) == 1, (stage, seen)
    assert not [n for n in seen if n in ("eg_project_fwd", "eg_tile_offsets", "eg_tile_emit", "eg_sort_pairs", "eg_composite_fwd", "eg_project_bwd")]
    if unit:
        assert seen.count("eg_composite_bwd_footprint_cams") == 1 and "eg_composite_bwd_footprint" not in seen


@pytest.mark.parametrize("strategy,mode", [("weighted", "antialiased"), ("whole", "antialiased"),
                                           ("bg_edge_ratio", "antialiased"), ("weighted", "classic")])
def test_rasterization_matches_oracle(env, strategy, mode):
    """The drop-in operator against the dense PyTorch oracle (autograd backward): every output of the call."""
    _lib, synth, O = env
    sc, fw, border, w, removed = strict_inputs(_scene(synth, n=3000), 0, strategy, seed=5)
    if mode == "classic":  # no compensation: different opacities, so its own borderline pixels
        from oracle import c_oracle as CO
        fw = CO.rasterize(sc.means.numpy(), sc.quats.numpy(), torch.exp(sc.log_scales).numpy(),
                          torch.sigmoid(sc.logit_opacities).squeeze(-1).numpy(), np.ones((sc.means.shape[0], 1), np.float32),
                          sc.viewmats[0].numpy(), sc.Ks[0].numpy(), sc.width, sc.height, antialiased=False)
        border = borderline_pixel_mask(fw, sc.gt[0])
        w = masked_weights(synth.weight_map(strategy, sc.gt[0], 1.0, torch.Generator().manual_seed(5)), border)
    cpu, gpu = _run_pair(env, sc, view=0, mode=mode, loss_fn=_l1_loss(sc, w, 0))
    io, ig = cpu["info"], gpu["info"]
    # integer outputs: the scene holds no Gaussian whose radius / cull / tile box hinges on rounding
    assert np.array_equal(to_np(io["radii"]), to_np(ig["radii"]))
    assert np.array_equal(to_np(io["tiles_per_gauss"]), to_np(ig["tiles_per_gauss"]))
    assert np.array_equal(to_np(io["isect_offsets"]), to_np(ig["isect_offsets"]))
    fo, fg = to_np(io["flatten_ids"]), to_np(ig["flatten_ids"])
    # depths are computed with a different operation order (fused multiply-add) on the device: two Gaussians of one
    # tile whose depths agree to the last bit may swap places.  Everything but such swaps must be identical.
    diff = fo != fg
    if diff.any():
        dep = to_np(io["depths"])[0]
        assert np.allclose(dep[fo[diff]], dep[fg[diff]], rtol=3e-7, atol=0), "sorted lists differ beyond depth ulps"
    # pixel indexing: the Gaussian that contributed last, on every pixel outside the borderline set
    ok = ~border
    lo = torch.from_numpy(fo)[io["last_ids"][0].long()]
    lg = torch.from_numpy(fg)[ig["last_ids"][0].cpu().long()]
    touched = (cpu["alpha"][0, ..., 0] > 0)
    n_idx = int((lo[ok & touched] != lg[ok & touched]).sum())
    assert n_idx == 0, f"{n_idx} pixels disagree on the last contributing Gaussian"
    # float outputs
    assert cpu["render"].shape == gpu["render"].shape == (1, sc.height, sc.width, 3)
    e = {"render": rel_err(gpu["render"][0][ok], cpu["render"][0][ok]), "alpha": rel_err(gpu["alpha"][0][ok], cpu["alpha"][0][ok])}
    assert_close(gpu["render"][0][ok], cpu["render"][0][ok], name="render")
    assert_close(gpu["alpha"][0][ok], cpu["alpha"][0][ok], name="alpha")
    # INSIDE the borderline pixel set the outcome is bounded, not free: the two sides may disagree on ONE decision of the
    # walk -- a Gaussian at alpha = 1/255 counted or not (|d alpha_pixel| <= T / 255), the walk ending at T = 1e-4 or one
    # contributor later (<= 1e-4), the 0.999 cap (<= 1e-3 T) -- so the pixel differs by at most 1 / 255 + 1e-4 + 1e-4
    if bool(border.any()):
        d_in = (gpu["alpha"][0, ..., 0].cpu()[border] - cpu["alpha"][0, ..., 0][border]).abs().max()
        e["alpha_diff_inside_borderline"] = float(d_in)
        assert float(d_in) <= 1.0 / 255.0 + 2e-4, float(d_in)
    assert abs(float(gpu["loss"]) - float(cpu["loss"])) <= 1e-4 * abs(float(cpu["loss"]))
    for k in ("means", "q", "ls", "lo"):
        e[f"grad {k}"] = rel_err(gpu[k].grad, cpu[k].grad)
        assert_close(gpu[k].grad, cpu[k].grad, name=f"grad {k}")
    e["v_means2d"] = rel_err(ig["means2d"].grad, io["means2d"].grad)
    e["absgrad"] = rel_err(ig["means2d"].absgrad, io["means2d"].absgrad)
    assert_close(ig["means2d"].grad, io["means2d"].grad, name="v_means2d")
    assert ig["means2d"].absgrad.shape == (1, sc.means.shape[0], 2)
    assert_close(ig["means2d"].absgrad, io["means2d"].absgrad, name="absgrad")
    record("rasterization_vs_torch_oracle", strategy=strategy, mode=mode, removed_borderline_gaussians=removed,
           borderline_pixels=int(border.sum()), pixels=int(border.numel()), max_rel_err=e,
           last_contributor_mismatches=n_idx, depth_ulp_swaps=int(diff.sum()))


def test_rasterization_general_colors(env):
    _lib, synth, O = env
    sc, removed = clean_scene(_scene(synth, n=1500, w=96, h=80), [0])
    n = sc.means.shape[0]
    colors = torch.rand(n, 3, generator=torch.Generator().manual_seed(9))
    keep = (~borderline_pixel_mask(oracle_forward(sc, 0))).float()  # borderline pixels: zero upstream gradient
    wr = torch.rand(80, 96, 3, generator=torch.Generator().manual_seed(10)) * keep[..., None]

    def fn(render, alpha, dev):
        return (render[0] * wr.to(dev)).sum() * 1e-3 + ((alpha[0, ..., 0] ** 2) * keep.to(dev)).sum() * 1e-3
    cpu, gpu = _run_pair(env, sc, colors=colors, loss_fn=fn)
    ok = keep.bool()
    assert_close(gpu["render"][0][ok], cpu["render"][0][ok], name="render")
    e = {}
    for k in ("means", "q", "ls", "lo", "col"):
        e[k] = rel_err(gpu[k].grad, cpu[k].grad)
        assert_close(gpu[k].grad, cpu[k].grad, name=f"grad {k}")
    e["absgrad"] = rel_err(gpu["info"]["means2d"].absgrad, cpu["info"]["means2d"].absgrad)
    assert_close(gpu["info"]["means2d"].absgrad, cpu["info"]["means2d"].absgrad, name="absgrad")
    record("general_colors_vs_torch_oracle", removed_borderline_gaussians=removed, borderline_pixels=int((~ok).sum()),
           max_rel_err=e)


def test_boundary_protocol(env, golden_dir):
    """The exact payload the reference's model class passes / reads back (boundary_trace.json,
    recorded by tests/golden/make_golden.py from edge_gs.py:250-275,607-613)."""
    import json
    import os
    _lib, synth, O = env
    from gsplat import rasterization  # the name the reference imports (edge_gs.py:8)
    tr = json.load(open(os.path.join(golden_dir, "boundary_trace.json")))
    kw = tr["kwargs"]
    cams = np.load(os.path.join(golden_dir, "cameras_00004926.npz"))
    N = kw["means"]["shape"][0]
    g = torch.Generator().manual_seed(1)
    args = dict(
        means=(1.1 * torch.rand(N, 3, generator=g) - 0.05).cuda().requires_grad_(True),
        quats=synth.random_quats(N, g).cuda().requires_grad_(True),
        scales=torch.full((N, 3), 0.004).cuda().requires_grad_(True),
        opacities=torch.full((N,), 0.08).cuda().requires_grad_(True),
        colors=torch.ones(N, 3).cuda(),
        viewmats=torch.from_numpy(cams["viewmats"][:1]).cuda(), Ks=torch.from_numpy(cams["Ks"][:1]).cuda())
    for k, v in kw.items():
        if isinstance(v, dict):
            assert list(args[k].shape) == v["shape"] and args[k].requires_grad == v["requires_grad"]
        else:
            args[k] = v
    render, alpha, info = rasterization(**args)
    assert list(render.shape) == [1, kw["height"], kw["width"], 3] and list(alpha.shape) == [1, kw["height"], kw["width"], 1]
    info["means2d"].retain_grad()
    xys, radii = info["means2d"], info["radii"][0]
    assert list(radii.shape) == tr["reads_back"]["info['radii'][0]"] and radii.dtype == torch.int32
    torch.clamp(render[0, ..., :3], 0, 1)[:, :, 0].mean().backward()
    absgrads = torch.zeros(N, device="cuda") + xys.absgrad[0].norm(dim=-1)  # update_absgrads
    assert absgrads.shape == (N,) and float(absgrads.max()) > 0
    assert args["means"].grad is not None and float(args["means"].grad.abs().max()) > 0


# ------------------------------------------------------------------ fused step
def test_adam_matches_torch(env):
    _lib, synth, O = env
    from edgegaussians_amd._lib import AdamHyper, call, ptr, stream
    N = 1000
    g = torch.Generator().manual_seed(2)
    shapes = {"means": 3, "scales": 3, "quats": 4, "opacities": 1}
    lrs = {"means": 2e-3, "scales": 1e-4, "quats": 1e-3, "opacities": 0.03}
    ps = {k: torch.randn(N, d, generator=g) for k, d in shapes.items()}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in ps.items()}
    opts = {k: torch.optim.Adam([ref[k]], lr=lrs[k]) for k in ref}
    dev = {k: v.clone().cuda() for k, v in ps.items()}
    m = torch.zeros(11 * N, device="cuda")
    v = torch.zeros(11 * N, device="cuda")
    for step in range(1, 6):
        grads = {k: torch.randn(N, d, generator=g) * 10 ** float(torch.randint(-6, 0, (1,), generator=g))
                 for k, d in shapes.items()}
        if step == 3:
            grads["quats"].zero_()  # zero gradients still move the parameter through the moments
        for k in ref:
            ref[k].grad = grads[k].clone()
            opts[k].step()
        gd = {k: t.cuda().contiguous() for k, t in grads.items()}
        h = AdamHyper(lrs["means"], lrs["scales"], lrs["quats"], lrs["opacities"], 0.9, 0.999, 1e-8, step)
        call("eg_adam_multi", ptr(dev["means"]), ptr(dev["scales"]), ptr(dev["quats"]), ptr(dev["opacities"]),
             ptr(gd["means"]), ptr(gd["scales"]), ptr(gd["quats"]), ptr(gd["opacities"]), ptr(m), ptr(v), N, h,
             None, None, stream())
    for k in ref:
        assert_close(dev[k], ref[k].data, rtol=1e-5, name=f"adam {k}")


def _reference_steps_cpu(O, synth, sc, views, strategies, lrs, n_steps):
    """K iterations of the reference protocol (train_gaussians.py:81-106) on the CPU oracle."""
    N = sc.means.shape[0]
    P = {"means": torch.nn.Parameter(sc.means.clone()), "scales": torch.nn.Parameter(sc.log_scales.clone()),
         "quats": torch.nn.Parameter(sc.quats.clone()), "opacities": torch.nn.Parameter(sc.logit_opacities.clone())}
    opts = {k: torch.optim.Adam([P[k]], lr=lrs[k]) for k in P}
    absgrads = torch.zeros(N)
    losses = []
    for s in range(n_steps):
        v = views[s]
        render, alpha, info = O.rasterization(
            means=P["means"], quats=P["quats"], scales=torch.exp(P["scales"]),
            opacities=torch.sigmoid(P["opacities"]).squeeze(-1), colors=torch.ones(N, 3),
            viewmats=sc.viewmats[v:v + 1], Ks=sc.Ks[v:v + 1], width=sc.width, height=sc.height, tile_size=16,
            packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        w = synth.weight_map(strategies[s], sc.gt[v], generator=torch.Generator().manual_seed(100 + s))
        loss = O.edge_step_loss(render[0, ..., 0], sc.gt[v], w)
        losses.append(float(loss))
        loss.backward()
        absgrads += info["means2d"].absgrad[0].norm(dim=-1)
        for o in opts.values():
            o.step()
            o.zero_grad()
    return P, absgrads, losses


def test_fused_train_step_matches_reference_protocol(env):
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = _scene(synth, n=2000, w=128, h=96, views=3)
    views = [0, 2, 1, 0]
    strategies = ["weighted", "bg_edge_ratio", "whole", "weighted"]
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    lrs = sched.at(0)
    P, absgrads, losses = _reference_steps_cpu(O, synth, sc, views, strategies, lrs, len(views))
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,
                     sc.width, sc.height, schedule=sched)
    tr.ensure_capacity()
    got = []
    for s, v in enumerate(views):
        w = synth.weight_map(strategies[s], sc.gt[v], generator=torch.Generator().manual_seed(100 + s)).cuda()
        tr.train_step(v, w)
        got.append(tr.pop_loss())
    assert not tr.overflowed()
    for a, b in zip(got, losses):
        assert abs(a - b) <= 2e-4 * abs(b), (got, losses)
    # Adam normalises the update, so the first steps move every parameter by ~lr regardless of the
    # gradient scale: compare the parameter DELTAS with a tolerance that admits sign-borderline noise
    for name, mine in (("means", tr.means), ("scales", tr.log_scales), ("quats", tr.quats),
                       ("opacities", tr.logit_opacities.view(-1, 1))):
        init = {"means": sc.means, "scales": sc.log_scales, "quats": sc.quats, "opacities": sc.logit_opacities}[name]
        d_ref = P[name].data - init
        d_got = mine.cpu() - init
        assert_close(d_got, d_ref, rtol=2e-3, max_bad=2e-2, name=f"delta {name}")
    assert_close(tr.absgrads, absgrads, max_bad=5e-3, name="absgrads")
    assert tr.absgrads_normalize_factor == 1 + len(views)


@pytest.mark.parametrize("segmented", [False, True])
def test_grad_step_equals_autograd_path(env, segmented):
    """eg_train_step without Adam (the data-parallel leg) == rasterization() + torch autograd, for both
    binning layouts of the step (count / scan / emit, and the one-pass segmented layout)."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, rasterization
    sc, _fw, _border, w, _ = strict_inputs(_scene(synth, n=2500, w=160, h=112, views=2), 1, "weighted")
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,
                     sc.width, sc.height, segmented=segmented)
    tr.ensure_capacity()
    w = w.cuda()
    tr.grad_step(1, w)
    gm, gq, gs, go = [t.clone() for t in tr.grad_views()]
    N = sc.means.shape[0]
    means = sc.means.clone().cuda().requires_grad_(True)
    ls = sc.log_scales.clone().cuda().requires_grad_(True)
    q = sc.quats.clone().cuda().requires_grad_(True)
    lo = sc.logit_opacities.clone().cuda().requires_grad_(True)
    render, alpha, info = rasterization(means, q, torch.exp(ls), torch.sigmoid(lo).squeeze(-1),
                                        torch.ones(N, 3, device="cuda"), sc.viewmats[1:2].cuda(), sc.Ks[1:2].cuda(),
                                        sc.width, sc.height, packed=False, absgrad=True, rasterize_mode="antialiased")
    loss = (w * (torch.clamp(render[0, ..., 0], 0, 1) - sc.gt[1].cuda()).abs()).sum()
    loss.backward()
    assert abs(tr.pop_loss() - float(loss)) <= 1e-5 * abs(float(loss))
    assert_close(gm, means.grad, rtol=1e-4, name="means")
    assert_close(gq, q.grad, rtol=1e-4, name="quats")
    assert_close(gs, ls.grad, rtol=1e-4, name="scales")
    assert_close(go, lo.grad.view(-1), rtol=1e-4, name="opacities")
    inc = tr.grads.view(-1)[11 * N:]
    assert_close(inc, info["means2d"].absgrad[0].norm(dim=-1), rtol=1e-4, name="absgrad inc")
    # the step does not materialise the images unless asked to; when asked they are the operator's
    assert tr.render is None and tr.alphas is None and tr.last_ids is None and tr.vpix is None
    tk = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,
                     sc.width, sc.height, keep_images=True, segmented=segmented)
    tk.ensure_capacity()
    tk.grad_step(1, w)
    assert_close(tk.render, render[0, ..., 0].detach(), rtol=1e-5, name="kept render")
    assert_close(tk.alphas, alpha[0, ..., 0].detach(), rtol=1e-5, name="kept alpha")
    # (with images the forward runs the workgroup-per-item kernels of the general API, without them the wave-autonomous
    # kernel: the same arithmetic in a different association, alphas agree to rounding)
    assert_close(tk.grad_views()[0], gm, rtol=1e-5, name="same gradients with and without images")


def _grad_step_vs_torch_oracle(env, sc, view, label, strategy="whole"):
    """fused eg_train_step (no Adam) against the dense PyTorch oracle's AUTOGRAD (tests/util.py)."""
    from tests.util import check_grad_step_vs_torch_oracle
    return check_grad_step_vs_torch_oracle(sc, view, label, strategy)


def test_fused_backward_big_footprints(env):
    """Footprints of thousands of pixels (one Gaussian takes most lanes of its wavefront)."""
    _lib, synth, O = env
    sc = synth.make_scene(200, 2, 320, 256, seed=4, spread_opacity=True, scale=0.12, anisotropy=3.0)
    _grad_step_vs_torch_oracle(env, sc, 0, "big_footprints")


def test_fused_step_with_transmittance_stops(env):
    """Opaque, heavily overlapping Gaussians: a share of the pixels hits the T <= 1e-4 stop, which is the only
    order-dependent part of the unit-colour path (slice re-walk in the forward, last-contributor
    test in the footprint backward)."""
    _lib, synth, O = env
    sc = synth.make_scene(4000, 2, 128, 96, seed=6, spread_opacity=False, scale=0.03, anisotropy=2.0)
    sc.logit_opacities[:] = torch.logit(torch.tensor(0.97))
    stopped = _grad_step_vs_torch_oracle(env, sc, 1, "transmittance_stops")
    assert stopped > 0.03, "scene must saturate a share of the pixels"


def test_fused_step_small_scene_vs_c_oracle(env):
    """gradients + one whole step against the sequential C oracle on the standard small scene (both views)."""
    _lib, synth, O = env
    for view, strategy in ((0, "weighted"), (1, "bg_edge_ratio")):
        check_fused_step_vs_c_oracle(_scene(synth, n=3000), view, strategy, f"small_scene_view{view}")


def test_native_data_parallel_run_equals_the_python_driver_single_rank_rccl(env):
    """eg_train_steps_dp: K view-sharded optimizer steps by ONE native call -- per step grad ->
    ncclAllReduce on the launch stream (a communicator of the library's own, created from a ncclUniqueId: here one rank)
    -> Adam + projection of the next view -- against dist.DataParallelStep.step driven from Python step by step through
    the same kernels and torch.distributed's collective: parameters, moments and absgrads torch.equal.  Also: a native
    run interrupted by a regulariser-free read-back and continued, the overflow journal holding its steps."""
    import os
    import socket
    import torch.distributed as dist
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    from edgegaussians_amd import dist as egdist
    sc = _scene(synth, n=2500, w=160, h=112, views=3)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    a, b = mk(), mk()
    a.ensure_capacity()
    b.ensure_capacity()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert egdist.init_native_comm() == 1 and _lib.load().eg_dp_world() == 1
        assert _lib.load().eg_dp_comm_count() == 1  # RCCL's own word for the communicator's size (ncclCommCount)
        # the run skips the identity sum of a one-rank communicator; the switch makes the [12 N] ncclAllReduce -- THE line
        # of the N-rank run -- execute here (round 5)
        assert _lib.load().eg_dp_force_all_reduce(1) == 0 and _lib.load().eg_dp_grad_all_reduces(None) == 0
        dpa, dpb = egdist.DataParallelStep(a), egdist.DataParallelStep(b)
        dpb.world = 2  # (the Python driver issues its collective even with one rank)
        assert dpa.native_ready()
        order = [0, 2, 1, 1, 0, 2, 2]
        wm = [synth.weight_map("weighted" if s != 1 else "bg_edge_ratio", sc.gt[v],
                               generator=torch.Generator().manual_seed(s)).cuda() for s, v in enumerate(order)]
        dpa.steps(order[:4], wm[:4], next_view=order[4])
        assert len(a._journal) == 4 and a._projected == order[4]
        la = a.pop_loss()                      # a read-back in the middle of the epoch: the projection is dropped
        dpa.steps(order[4:], wm[4:])
        for s, v in enumerate(order):
            if s == 4:
                lb = b.pop_loss()
            dpb.step(v, wm[s], next_view=order[s + 1] if s + 1 < len(order) else None)
        torch.cuda.synchronize()
        assert abs(la - lb) <= 1e-6 * abs(lb)
        import ctypes as _C
        fl = _C.c_int64()
        assert _lib.load().eg_dp_grad_all_reduces(_C.byref(fl)) == len(order) and fl.value == len(order) * 12 * a.N
        # ... timed by events the native run records around it on the launch stream
        assert _lib.load().eg_dp_comm_timing_begin(2) == 0
        c = mk()
        c.ensure_capacity()
        dpc = egdist.DataParallelStep(c)
        dpc.steps(order[:3], wm[:3])
        mean_us, max_us, cnt = _C.c_float(), _C.c_float(), _C.c_int32()
        assert _lib.load().eg_dp_comm_timing_end(_C.byref(mean_us), _C.byref(max_us), _C.byref(cnt)) == 0
        assert cnt.value == 2 and 0.0 < mean_us.value <= max_us.value < 1e6
        assert _lib.load().eg_dp_grad_all_reduces(None) == len(order) + 3
        # the small collective that rides the same communicator
        t = torch.arange(5, dtype=torch.float32, device="cuda")
        from edgegaussians_amd._lib import call, ptr, stream
        call("eg_dp_all_reduce", ptr(t), 5, stream())
        assert torch.equal(t.cpu(), torch.arange(5, dtype=torch.float32))
    finally:
        _lib.load().eg_dp_shutdown()
        dist.destroy_process_group()
    for x, y, name in ((a.means, b.means, "means"), (a.log_scales, b.log_scales, "scales"), (a.quats, b.quats, "quats"),
                       (a.logit_opacities, b.logit_opacities, "opac"), (a.absgrads, b.absgrads, "absgrads"),
                       (a.adam_m, b.adam_m, "m"), (a.adam_v, b.adam_v, "v")):
        assert torch.equal(x, y), name
    assert a.step == b.step == len(order) and a.group_steps == b.group_steps
    assert abs(a.pop_loss() - b.pop_loss()) <= 1e-6


def test_data_parallel_leg_on_gpu_single_rank(env):
    """The multi-GPU code path (grad_step -> RCCL all-reduce of the fused [N,12] buffer -> eg_adam_multi)
    on one rank must reproduce the fused single-GPU step: same gradients, same Adam arithmetic."""
    import os
    import socket
    import torch.distributed as dist
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    from edgegaussians_amd import dist as egdist
    sc = _scene(synth, n=2500, w=160, h=112, views=3)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    a, b, c = mk(), mk(), mk()
    a.ensure_capacity()
    b.ensure_capacity()
    c.ensure_capacity()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dp = egdist.DataParallelStep(b)
        dp.world = 2  # force the collective even with one rank
        dpc = egdist.DataParallelStep(c)  # ... and with the next view announced: Adam + its projection in one launch
        dpc.world = 2
        order = [0, 2, 1, 1, 0]
        for s, v in enumerate(order):
            w = synth.weight_map("weighted" if s != 1 else "bg_edge_ratio", sc.gt[v],
                                 generator=torch.Generator().manual_seed(s)).cuda()
            a.train_step(v, w)
            dp.step(v, w)
            # (step 2 announces a view that is NOT the one taken next: the pre-projection must then be ignored)
            nxt = None if s + 1 >= len(order) else (order[s + 1] if s != 2 else 0)
            dpc.step(v, w, next_view=nxt)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    for x, y, name in ((a.means, b.means, "means"), (a.log_scales, b.log_scales, "scales"), (a.quats, b.quats, "quats"),
                       (a.logit_opacities, b.logit_opacities, "opac"), (a.absgrads, b.absgrads, "absgrads"),
                       (a.adam_m, b.adam_m, "m"), (a.adam_v, b.adam_v, "v")):
        assert_close(x, y, rtol=1e-6, name=name)
    for x, y, name in ((b.means, c.means, "means"), (b.log_scales, c.log_scales, "scales"), (b.quats, c.quats, "quats"),
                       (b.logit_opacities, c.logit_opacities, "opac"), (b.absgrads, c.absgrads, "absgrads"),
                       (b.adam_m, c.adam_m, "m"), (b.adam_v, c.adam_v, "v")):
        assert_close(x, y, rtol=1e-6, name="announced next view: " + name)
    assert a.absgrads_normalize_factor == b.absgrads_normalize_factor == c.absgrads_normalize_factor == 6
    la, lb, lc = a.pop_loss(), b.pop_loss(), c.pop_loss()
    assert abs(la - lb) < 1e-6 and abs(lb - lc) < 1e-6


# ------------------------------------------------------------------ edge cases and full-size properties
def test_empty_and_degenerate_scenes(env):
    """M = 0 (everything culled), N = 1, and an image smaller than one tile: no faults, exact zeros."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, rasterization
    sc = _scene(synth, n=64, w=40, h=24, views=1)
    R, t = sc.viewmats[0, :3, :3], sc.viewmats[0, :3, 3]
    behind = (-R.T @ t) - 3.0 * R[2]  # a point behind the camera
    means = (behind[None] + 0.01 * torch.randn(64, 3)).cuda().requires_grad_(True)
    args = dict(quats=sc.quats.cuda(), scales=torch.exp(sc.log_scales).cuda(),
                opacities=torch.sigmoid(sc.logit_opacities).squeeze(-1).cuda(), colors=torch.ones(64, 3).cuda(),
                viewmats=sc.viewmats[:1].cuda(), Ks=sc.Ks[:1].cuda(), width=40, height=24, packed=False,
                absgrad=True, rasterize_mode="antialiased")
    r, a, info = rasterization(means=means, **args)
    assert float(r.abs().max()) == 0 and float(a.abs().max()) == 0 and int(info["radii"].abs().sum()) == 0
    assert info["flatten_ids"].numel() == 0 and int(info["isect_offsets"].abs().sum()) == 0
    (r.sum() + a.sum()).backward()
    assert float(means.grad.abs().max()) == 0
    # fused step on the same empty view: loss = sum w |0 - gt|, parameters move only through Adam's zero grads
    tr = EdgeTrainer(means.detach().cpu(), sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, 40, 24)
    w = synth.weight_map("whole", sc.gt[0]).cuda()
    tr.train_step(0, w)
    assert abs(tr.pop_loss() - float((w.cpu() * sc.gt[0]).sum())) < 1e-6 and tr.last_m() == 0
    assert torch.equal(tr.means.cpu(), means.detach().cpu())
    # N = 1, image 9x7 (smaller than a tile)
    one = synth.make_scene(1, 1, 9, 7, seed=1, scale=0.3, spread_opacity=True)
    one.means[0] = torch.tensor([0.5, 0.5, 0.5])
    ok1 = ~borderline_pixel_mask(oracle_forward(one, 0))
    cpu, gpu = _run_pair(env, one, loss_fn=lambda render, alpha, dev: ((render[0, ..., 0] ** 2) * ok1.to(dev)).sum())
    assert_close(gpu["render"][0][ok1], cpu["render"][0][ok1], name="render 1-gaussian")
    assert_close(gpu["means"].grad, cpu["means"].grad, rtol=2e-4, name="grad 1-gaussian")


def test_full_size_properties_config2(env):
    """BASELINE config 2 (100 k Gaussians, 512x512): the size-independent properties of the path, and the fused step
    against the drop-in operator.  (The headline scene itself -- real poses, both opacity regimes -- is compared with
    the C oracle in tests/test_gpu_fullsize.py::test_fused_step_vs_c_oracle_config2_real_poses.)"""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, rasterization
    W, H = 512, 512
    sc, _fw, _border, w_strict, _removed = strict_inputs(
        synth.make_scene(100_000, 2, W, H, seed=0, anisotropy=5.0, spread_opacity=True), 0, "weighted")
    n = sc.means.shape[0]
    dev = "cuda"
    p = [t.clone().to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
    r, a, info = rasterization(p[0], p[1], torch.exp(p[2]), torch.sigmoid(p[3]).squeeze(-1), torch.ones(n, 3, device=dev),
                               sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), W, H, packed=False, absgrad=True,
                               rasterize_mode="antialiased")
    # (1) unit colours: every channel == accumulated alpha, all in [0, 1)
    assert float((r[..., 0] - a[..., 0]).abs().max()) < 1e-6 and float(a.max()) < 1.0 and float(a.min()) >= 0
    # (2) binning bookkeeping: M = sum tiles_per_gauss, offsets monotone, every tile segment sorted by
    #     (depth bits, id), isect ids carry the tile of their segment
    ids, flat, offs = info["isect_ids"], info["flatten_ids"], info["isect_offsets"].reshape(-1)
    M = int(info["tiles_per_gauss"].sum())
    assert ids.numel() == flat.numel() == M and bool((offs[1:] >= offs[:-1]).all()) and int(offs[-1]) <= M
    assert bool((ids[1:] >= ids[:-1]).all())  # globally sorted by (tile, depth)
    tile_of = torch.bucketize(torch.arange(M, device=dev), offs.long(), right=True) - 1
    assert torch.equal((ids >> 32), tile_of)
    depth_bits = info["depths"][0].view(torch.int32).long()[flat.long()]
    assert torch.equal(ids & 0xffffffff, depth_bits)
    same = (ids[1:] == ids[:-1])
    assert bool((flat[1:][same] > flat[:-1][same]).all())  # ties: ascending Gaussian id (stable)
    # (3) fused path == operator path on the same inputs: loss and gradients
    w = w_strict.to(dev)
    loss = (w * (torch.clamp(r[0, ..., 0], 0, 1) - sc.gt[0].to(dev)).abs()).sum()
    loss.backward()
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H)
    tr.ensure_capacity()
    tr.grad_step(0, w)
    assert abs(tr.pop_loss() - float(loss)) <= 2e-5 * abs(float(loss)) and not tr.overflowed()
    assert tr.last_m() <= M  # tight tile boxes only ever drop (Gaussian, tile) pairs
    gm, gq, gs, go = tr.grad_views()
    for got, want, name in ((gm, p[0].grad, "means"), (gq, p[1].grad, "quats"), (gs, p[2].grad, "scales"),
                            (go, p[3].grad.view(-1), "opac")):
        assert_close(got, want, rtol=1e-4, name=name)
    assert_close(tr.grads.view(-1)[11 * n:], info["means2d"].absgrad[0].norm(dim=-1), rtol=1e-4, name="absgrad")
    # (4) determinism of everything that is not a float atomic: two forwards are bit-identical
    r2, a2, info2 = rasterization(p[0].detach(), p[1].detach(), torch.exp(p[2]).detach(), torch.sigmoid(p[3]).squeeze(-1).detach(),
                                  torch.ones(n, 3, device=dev), sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), W, H,
                                  packed=False, rasterize_mode="antialiased")
    assert torch.equal(r2, r.detach()) and torch.equal(info2["flatten_ids"], flat)


@pytest.mark.parametrize("W,H", [(2100, 2100), (1900, 1600)])
def test_large_tile_grids_take_the_fallback_binning_paths(env, W, H):
    """T > 16384 tiles: the LDS-privatised counters no longer fit and projection / count / emit fall back
    to direct atomics; 8192 < T <= 16384: projection keeps LDS counters, emit falls back.  Fused step vs the
    plain-C oracle."""
    _lib, synth, O = env
    T = math.ceil(W / 16) * math.ceil(H / 16)
    assert T > 8192
    sc = synth.make_scene(6000, 1, W, H, seed=2, spread_opacity=True, scale=0.006)
    tr, _, _ = check_fused_step_vs_c_oracle(sc, 0, "weighted", f"large_grid_{W}x{H}")
    assert int(tr.tile_counts.abs().sum()) == 0  # emit returned every counter to zero


# ------------------------------------------------------------------ 8(f): kNN + orientation regularisers
def _knn(R, pts, k, method, **kw):
    """method: "exhaustive" (eg_knn_small), "grid" (eg_knn_auto: grid chosen on the device), "hostgrid" (eg_knn)."""
    if method == "hostgrid":
        return R.knn(pts, k, method="grid", grid=R.make_grid(pts, margin=0.05), **kw)
    return R.knn(pts, k, method=method, **kw)


@pytest.mark.parametrize("method", ["grid", "hostgrid", "exhaustive"])
@pytest.mark.parametrize("k,clustered", [(6, False), (6, True), (16, False), (21, True)])
def test_knn_matches_sklearn(env, k, clustered, method):
    """The three entries (grid chosen on the device: eg_knn_auto, the default above 5 k points; caller's grid:
    eg_knn; exhaustive: eg_knn_small) against sklearn's KD-tree, which is what the reference calls
    (edge_gs.py:135-151)."""
    from sklearn.neighbors import NearestNeighbors
    from edgegaussians_amd import regularizers as R
    g = torch.Generator().manual_seed(11)
    n = 6000
    if clustered:  # points along a few line segments (what trained edge Gaussians look like) + noise
        t = torch.rand(n, 1, generator=g)
        seg = torch.randint(0, 6, (n,), generator=g)
        a, b = torch.rand(6, 3, generator=g), torch.rand(6, 3, generator=g)
        pts = a[seg] * (1 - t) + b[seg] * t + 0.003 * torch.randn(n, 3, generator=g)
    else:
        pts = torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.6, 0.3])
    if clustered:  # ... and faint floaters spread through the volume (three quarters of a trained ABC model)
        pts[n // 2:] = torch.rand(n - n // 2, 3, generator=g) * 1.3 - 0.15
    idx, dist = _knn(R, pts.cuda(), k, method, want_dist=True)
    d_ref, i_ref = NearestNeighbors(n_neighbors=k + 1, algorithm="auto", metric="euclidean").fit(pts.numpy()).kneighbors(pts.numpy())
    d_ref, i_ref = d_ref[:, 1:], i_ref[:, 1:]  # k_nearest_sklearn drops the point itself (edge_gs.py:151)
    assert np.allclose(to_np(dist), d_ref, rtol=1e-4, atol=1e-7)
    assert (to_np(idx) == i_ref).mean() > 0.999  # equal up to exact distance ties
    # the reference's neighbour set: ranks 2 .. k+1
    nn = R.reference_nn_indices(pts.cuda(), k - 1, method=method.replace("hostgrid", "grid"))
    assert nn.shape == (n, k - 1) and (to_np(nn) == i_ref[:, 1:]).mean() > 0.999


def test_knn_exhaustive_equals_grid_search(env):
    """The searches order candidates by the same (distance, index) key: identical tables from the exhaustive
    search, the device-chosen grid and a caller-chosen grid, from one point up to 10^5, with coincident points,
    and with all points on a line / in one spot (degenerate bounding boxes)."""
    from edgegaussians_amd import regularizers as R
    g = torch.Generator().manual_seed(5)
    for n, k in ((1, 3), (2, 1), (7, 6), (63, 8), (300, 6), (5000, 11), (40000, 6), (100000, 21), (3000, 4), (3001, 4)):
        pts = torch.rand(n, 3, generator=g)
        if n >= 300:
            pts[: n // 10] = pts[n // 10: 2 * (n // 10)]  # exact duplicates: distance ties resolved by index
        if n == 3000:
            pts[:, 1:] = 0.25  # all on one line
        if n == 3001:
            pts[:] = pts[0]    # all in one spot: every distance ties
        pts = pts.cuda()
        ia, da = R.knn(pts, k, want_dist=True, method="grid")
        ib, db = R.knn(pts, k, want_dist=True, method="exhaustive")
        ic, dc = _knn(R, pts, k, "hostgrid", want_dist=True)
        assert torch.equal(ia, ib) and torch.equal(ia, ic), (n, k, int((ia != ib).sum()), int((ia != ic).sum()))
        assert torch.allclose(da, db, rtol=1e-6, atol=0) and torch.allclose(da, dc, rtol=1e-6, atol=0)
        assert int((ia >= 0).sum(1).min()) == min(k, n - 1)  # fewer than k other points: the tail stays -1


def test_direction_and_ratio_losses_match_autograd(env):
    """Values and gradients against the reference's formulas (edge_gs.py:346-380) under autograd."""
    from edgegaussians_amd import regularizers as R
    g = torch.Generator().manual_seed(12)
    n, k = 3000, 5
    means = torch.rand(n, 3, generator=g)
    quats = torch.randn(n, 4, generator=g)
    ls = torch.log(0.004 * (1 + 4 * torch.rand(n, 3, generator=g)))
    nn = R.reference_nn_indices(means.cuda(), k)
    m, q, s = means.clone().requires_grad_(True), quats.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    # compute_direction_loss, enforce_full
    qn = torch.nn.functional.normalize(q, p=2, dim=1)
    w, x, y, z = qn.unbind(-1)
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)
    major = Rm[torch.arange(n), :, torch.argmax(torch.exp(s).abs(), dim=-1)]
    nd = m[:, None, :] - m[nn.cpu().long()]
    nd = nd / nd.norm(dim=-1, keepdim=True)
    loss_ref = 1.0 - (major[:, None, :] * nd).sum(-1).abs().mean(-1).mean()
    loss_ref.backward()
    loss, gm, gq = R.direction_loss(means.cuda(), quats.cuda(), ls.cuda(), nn)
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert_close(gm, m.grad, rtol=1e-4, max_bad=1e-3, name="dir dmeans")
    assert_close(gq, q.grad, rtol=1e-4, max_bad=1e-3, name="dir dquats")
    # compute_ratio_loss
    s2 = ls.clone().requires_grad_(True)
    srt, _ = torch.sort(torch.exp(s2), dim=-1, descending=True)
    r_ref = (srt[:, 1] / srt[:, 0]).mean()
    r_ref.backward()
    r, gs = R.ratio_loss(ls.cuda())
    assert abs(float(r) - float(r_ref)) < 1e-6
    assert_close(gs, s2.grad, rtol=1e-5, name="ratio dlogscales")


@pytest.mark.parametrize("search", ["grid", "exhaustive"])
@pytest.mark.parametrize("method,k", [("enforce_full", 5), ("enforce_full", 10), ("enforce_half", 5), ("enforce_half", 10)])
def test_regularisers_match_reference_functions(env, golden_dir, method, k, search):
    """eg_knn / eg_direction_loss / eg_ratio_loss against the REFERENCE's own update_nearest_neighbors,
    compute_direction_loss, compute_ratio_loss (edge_gs.py:326-380: sklearn KD-tree + torch autograd), run in
    the build container on a seeded trained-like state (tests/golden/make_golden.py:regularizers)."""
    import os
    from edgegaussians_amd import regularizers as R
    d = np.load(os.path.join(golden_dir, "regularizers.npz"))
    means, quats, ls = (torch.from_numpy(d[x]).cuda() for x in ("means", "quats", "log_scales"))
    tag = f"{method}_{k}"
    nn_ref = d[f"nn_{tag}"]
    nn = R.reference_nn_indices(means, k, method, method=search)
    assert nn.shape == nn_ref.shape == (means.shape[0], 2 * k if method == "enforce_half" else k)
    same_rows = (to_np(nn) == nn_ref).all(axis=1)
    # index work: exact, except rows in which two neighbours are equidistant to the last bit (KD-tree vs grid order)
    assert same_rows.mean() > 0.999, same_rows.mean()
    nn_use = torch.from_numpy(nn_ref).cuda()  # the reference's table: the loss comparison is then input-identical
    loss, gm, gq = R.direction_loss(means, quats, ls, nn_use, k if method == "enforce_half" else 0)
    e = {"loss": abs(float(loss) - float(d[f"dir_loss_{tag}"])) / abs(float(d[f"dir_loss_{tag}"])),
         "gmeans": rel_err(gm, d[f"dir_gmeans_{tag}"]), "gquats": rel_err(gq, d[f"dir_gquats_{tag}"])}
    assert e["loss"] < 1e-5
    assert_close(gm, d[f"dir_gmeans_{tag}"], rtol=1e-4, name="dir dmeans")
    assert_close(gq, d[f"dir_gquats_{tag}"], rtol=1e-4, name="dir dquats")
    r, gs = R.ratio_loss(ls)
    e["ratio"] = abs(float(r) - float(d["ratio_loss"])) / float(d["ratio_loss"])
    e["gscales"] = rel_err(gs, d["ratio_gscales"])
    assert e["ratio"] < 1e-5
    assert_close(gs, d["ratio_gscales"], rtol=1e-5, name="ratio dlogscales")
    record("regularisers_vs_reference_functions", method=method, k=k, search=search, nn_rows_identical=float(same_rows.mean()), max_rel_err=e)


def test_regulariser_step_advances_only_three_optimizers(env):
    """train_gaussians.py:108-131: the means / scales / quats optimizers step (their own step counts
    advance), the opacity optimizer does not; a later projection step uses the drifted counts."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    from edgegaussians_amd import regularizers as R
    sc = _scene(synth, n=1500, w=96, h=80, views=2)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    lrs = sched.at(0)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, 96, 80,
                     schedule=sched)
    # torch-side emulation with the reference's optimizers (zero_grad leaves ZERO tensors in torch 1.13)
    P = {"means": torch.nn.Parameter(sc.means.clone().cuda()), "scales": torch.nn.Parameter(sc.log_scales.clone().cuda()),
         "quats": torch.nn.Parameter(sc.quats.clone().cuda()), "opacities": torch.nn.Parameter(sc.logit_opacities.clone().cuda())}
    opts = {k: torch.optim.Adam([P[k]], lr=lrs[k]) for k in P}

    def torch_step(grads, names):
        for k in names:
            P[k].grad = grads.get(k, torch.zeros_like(P[k])).clone()
            opts[k].step()

    w = synth.weight_map("weighted", sc.gt[0]).cuda()
    # 1) a projection step through the autograd path gives the torch side its gradients
    from edgegaussians_amd import rasterization
    def proj_grads():
        p = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        r, _, info = rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]).squeeze(-1),
                                   torch.ones(1500, 3, device="cuda"), sc.viewmats[:1].cuda(), sc.Ks[:1].cuda(), 96, 80,
                                   packed=False, absgrad=True, rasterize_mode="antialiased")
        loss = (w * (torch.clamp(r[0, ..., 0], 0, 1) - sc.gt[0].cuda()).abs()).sum()
        loss.backward()
        return {k: v.grad for k, v in p.items()}, float(loss)
    gproj, l0 = proj_grads()
    torch_step(gproj, ["means", "scales", "quats", "opacities"])
    tr.train_step(0, w)
    # 2) direction step
    nn = R.reference_nn_indices(P["means"].data, 5)
    loss, dm, dq = R.direction_loss(P["means"].data, P["quats"].data, P["scales"].data, nn)
    lam = l0 * 0.01 / float(loss)
    torch_step({"means": dm * lam, "quats": dq * lam}, ["means", "scales", "quats"])
    v = tr.regulariser_step("direction", l0, 0.01)
    assert abs(v - float(loss)) < 1e-5
    # 3) ratio step
    loss, ds = R.ratio_loss(P["scales"].data)
    torch_step({"scales": ds * (l0 * 0.01 / float(loss))}, ["means", "scales", "quats"])
    tr.regulariser_step("ratio", l0, 0.01)
    assert tr.group_steps == [3, 3, 3, 1]
    # 4) another projection step: bias corrections now differ per optimizer
    gproj, _ = proj_grads()
    torch_step(gproj, ["means", "scales", "quats", "opacities"])
    tr.train_step(0, w)
    for name, mine in (("means", tr.means), ("scales", tr.log_scales), ("quats", tr.quats),
                       ("opacities", tr.logit_opacities.view(-1, 1))):
        init = {"means": sc.means, "scales": sc.log_scales, "quats": sc.quats, "opacities": sc.logit_opacities}[name]
        assert_close(mine.cpu() - init, P[name].data.cpu() - init, rtol=2e-3, max_bad=2e-2, name=f"delta {name}")


# ------------------------------------------------------------------ the reference's loop, end to end
def test_training_loop_on_real_edge_maps(env, golden_dir):
    """`train()` (train_gaussians.py:144-222) with the reference's own ABC config sections on the four
    real DexiNed views of scan 00004926 (fixtures): alternating loss strategies, LR schedule, both
    regularisers, duplicate / opacity-cull / not-projecting-cull events.  Checks that the loop runs
    through every branch, that N follows the calendar and that the projection loss falls."""
    import json
    import os
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, train
    cams = np.load(os.path.join(golden_dir, "cameras_00004926.npz"))
    edges = np.load(os.path.join(golden_dir, "edges_00004926.npz"))
    views = list(edges["views"])
    H, W = int(cams["height"]), int(cams["width"])
    gt = torch.zeros(len(views), H, W)
    for i, k in enumerate(views):
        gt[i].view(-1)[torch.from_numpy(edges[f"idx_{k}"]).long()] = torch.from_numpy(edges[f"val_{k}"]).float() / 255.0
    g = torch.Generator().manual_seed(0)
    n = 2500  # init_min_num_gaussians (configs/ABC_DexiNed.json:27)
    means = 1.1 * torch.rand(n, 3, generator=g) - 0.55 + 0.5
    tr = EdgeTrainer(means, torch.full((n, 3), math.log(0.004)), synth.random_quats(n, g),
                     torch.logit(torch.full((n, 1), 0.08)), torch.from_numpy(cams["viewmats"][views]),
                     torch.from_numpy(cams["Ks"][views]), gt, W, H)
    model_cfg = {"if_duplicate_high_pos_grad": True, "dup_threshold_type": "absolute", "dup_threshold_value": 0.5,
                 "dup_factor": 3, "dup_high_pos_grads_at_epoch": [4, 8], "init_dup_rand_noise_scale": 0.05,
                 "if_cull_low_opacity": True, "cull_opacity_type": "absolute", "cull_opacity_value": 0.05,
                 "cull_opacity_at_epoch": [10], "if_cull_gaussians_not_projecting": True,
                 "cull_gaussians_not_projecting_at_epoch": [12], "cull_gaussians_not_projecting_threshold": 0.1}
    optim = json.load(open(os.path.join(golden_dir, "abc_optim_config.json")))
    for k in ("scales", "opacities", "quats"):
        optim[k]["start_at_epoch"] = 2  # compress the calendar: 16 epochs instead of 400
    training_cfg = {"num_epochs": 16, "optim": optim, "loss": {
        "orientation_losses": {"start_dir_loss_at_epoch": 9, "start_ratio_loss_at_epoch": 6, "dir_loss_num_nn": 5,
                               "dir_loss_scale_factor": 0.01, "ratio_loss_scale_factor": 0.01},
        "projection_losses": {"lambda_annealing": "constant", "lambda_start": 1, "lambda_end": 1,
                              "loss_before_alternating": "whole", "less_freq_loss": "bg_edge_ratio",
                              "more_freq_loss": "whole", "start_alternating_at_epoch": 3,
                              "bg_edge_pixel_ratio_annealing": "constant", "bg_edge_pixel_ratio_start": 1,
                              "bg_edge_pixel_ratio_end": 1, "sampling_whole_num_epochs_ratio": 5}}}
    counts = []
    order = lambda epoch: [v for _ in range(12) for v in torch.randperm(4, generator=g).tolist()]  # noqa: E731
    hist = train(tr, model_cfg, training_cfg, order, on_epoch=lambda e, l, nn: counts.append(nn))
    assert len(hist) == 16 and all(math.isfinite(x) for x in hist) and not tr.overflowed()
    assert hist[3] < hist[0]  # 'whole' epochs: comparable losses, and training reduces them
    assert counts[3] == 2500 and counts[4] > counts[3] and counts[8] >= counts[7]  # duplication events
    assert counts[10] <= counts[9] and counts[12] <= counts[11]                   # cull events
    assert tr.group_steps[0] > tr.group_steps[3] > 0                              # regulariser steps happened
    assert torch.isfinite(tr.means).all() and torch.isfinite(tr.logit_opacities).all()
    sd = tr.state_dict()
    assert sd["gauss_params.opacities"].shape == (tr.N, 1)


# ------------------------------------------------------------------ densify / cull vs the reference's own outputs
def test_densify_cull_golden(env, golden_dir):
    import os
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    d = np.load(os.path.join(golden_dir, "densify_cull.npz"))
    cams = np.load(os.path.join(golden_dir, "cameras_00004926.npz"))
    edges = np.load(os.path.join(golden_dir, "edges_00004926.npz"))
    keep_views = list(edges["views"])
    H, W = int(cams["height"]), int(cams["width"])
    T = lambda k: torch.from_numpy(d[k])  # noqa: E731
    gt = torch.zeros(len(keep_views), H, W)
    for i, k in enumerate(keep_views):
        gt[i].view(-1)[torch.from_numpy(edges[f"idx_{k}"]).long()] = torch.from_numpy(edges[f"val_{k}"]).float() / 255.0
    tr = EdgeTrainer(T("before_means"), T("before_scales"), T("before_quats"), T("before_opacities"),
                     torch.from_numpy(cams["viewmats"][keep_views]), torch.from_numpy(cams["Ks"][keep_views]), gt, W, H)
    N = tr.N
    for t, key in ((tr.adam_m, "exp_avg"), (tr.adam_v, "exp_avg_sq")):
        off = 0
        for name, dim in (("means", 3), ("scales", 3), ("quats", 4), ("opacities", 1)):
            t[off:off + N * dim] = T(f"before_{name}_{key}").reshape(-1).cuda()
            off += N * dim
    tr.absgrads = T("absgrads").cuda()
    tr.absgrads_normalize_factor = float(d["absgrads_factor"])
    # duplicate_high_pos_gradients ('absolute', 0.5, dup_factor 3, noise 0.05: configs/ABC_DexiNed.json)
    n_sel = tr.duplicate_high_pos_gradients(0.5, 3, 0.05, noise=T("dup_noise"))
    assert tr.N == d["dup_means"].shape[0] and n_sel * 2 == tr.N - N
    for name, p in tr._params().items():
        assert_close(p, T(f"dup_{name}"), rtol=1e-6, name=f"dup {name}")
        assert_close(tr._moment_views(tr.adam_m)[name], T(f"dup_{name}_exp_avg"), rtol=1e-6, name=f"dup m {name}")
        assert_close(tr._moment_views(tr.adam_v)[name], T(f"dup_{name}_exp_avg_sq"), rtol=1e-6, name=f"dup v {name}")
    assert tr.absgrads.shape[0] == int(d["dup_absgrads_len"]) and float(tr.absgrads.abs().sum()) == 0
    assert tr.absgrads_normalize_factor == float(d["dup_factor_after"])
    # cull_gaussians_opacity (absolute 0.05)
    tr.absgrads = torch.arange(tr.N, device="cuda").float()
    tr.cull_opacity(0.05)
    for name, p in tr._params().items():
        assert_close(p, T(f"cull_{name}"), rtol=1e-6, name=f"cull {name}")
        assert_close(tr._moment_views(tr.adam_m)[name], T(f"cull_{name}_exp_avg"), rtol=1e-6, name=f"cull m {name}")
    assert np.array_equal(to_np(tr.absgrads), d["cull_absgrads"])
    # cull_gaussians_not_projecting over the 4 fixture views
    assert_close(tr.means, T("np_means_before"), rtol=1e-6, name="np means")
    tr.absgrads = torch.arange(tr.N, device="cuda").float()
    tr.cull_not_projecting((gt >= 0.5).to(torch.uint8).cuda(), 0.1)
    assert np.array_equal(to_np(tr.absgrads).astype(np.int64), d["np_kept_index"])


def test_filter_by_projection_matches_reference(env, golden_dir):
    """Device twin of the edge-extraction filter (filtering.py:80-123) vs the reference's own inlier
    masks (fixture) and vs the oracle's per-Gaussian visibility: index work, so exact."""
    from tests.util import filter_fixture
    _lib, synth, O = env
    from edgegaussians_amd import filtering
    d, images, cameras = filter_fixture(golden_dir)
    for thr in (0.1, 0.3):
        got = filtering.filter_by_projection(d["means"], [torch.from_numpy(i) for i in images], cameras, thr)
        assert got.dtype == bool and got.shape == (d["means"].shape[0],)
        assert np.array_equal(got, d[f"inliers_{thr}"])
    assert filtering.filter_by_projection(d["means"][:0], images, cameras).shape == (0,)
    assert np.array_equal(filtering.filter_by_opacity(np.array([[0.2], [0.01]]), 0.05), np.array([True, False]))


def test_spatial_row_order_is_a_pure_relabelling(env):
    """EdgeTrainer(spatial_order=True) keeps its rows in Morton order internally; gradients, densify /
    cull events and checkpoints must be those of the as-given order once mapped back (ref_index)."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    sc = _scene(synth, n=3000, w=160, h=112, views=2)
    mk = lambda so: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks,  # noqa: E731
                                sc.gt, sc.width, sc.height, spatial_order=so)
    ta, tb = mk(False), mk(True)
    assert tb.ref_index is not None and not torch.equal(tb.ref_index, torch.arange(3000, device="cuda"))
    for k, v in ta.state_dict().items():
        assert torch.equal(v, tb.state_dict()[k]), k
    w = synth.weight_map("weighted", sc.gt[1]).cuda()
    ta.ensure_capacity(); tb.ensure_capacity()
    ta.grad_step(1, w); tb.grad_step(1, w)
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-5 * abs(la)
    for ga, gb, name in zip(ta.grad_views(), tb.grad_views(), ("means", "quats", "scales", "opacities")):
        assert_close(tb._in_reference_order(gb), ga, rtol=1e-4, name=f"grad {name}")
    # three optimizer steps, then the densify / cull events with the same (reference-order) noise
    for t in (ta, tb):
        for s in range(3):
            t.train_step(s % 2, w)
    assert_close(tb._in_reference_order(tb.absgrads), ta.absgrads, rtol=1e-4, max_bad=2e-3, name="absgrads")
    tb.absgrads = ta.absgrads[tb.ref_index].clone()          # identical selection on both sides
    for name in ("means", "log_scales", "quats", "logit_opacities"):
        setattr(tb, name, getattr(ta, name)[tb.ref_index].clone())
    ma, mb = ta._moment_views(ta.adam_m), tb._moment_views(tb.adam_m)
    for k in ma:
        mb[k].copy_(ma[k][tb.ref_index])
    g = ta.absgrads / ta.absgrads_normalize_factor
    n_sel = int(((g - g.min()) / (g.max() - g.min()) > 0.5).sum())
    assert n_sel > 0
    noise = torch.randn(2 * n_sel, 3, generator=torch.Generator().manual_seed(3))
    assert ta.duplicate_high_pos_gradients(0.5, 3, 0.05, noise.clone()) == n_sel
    assert tb.duplicate_high_pos_gradients(0.5, 3, 0.05, noise.clone()) == n_sel
    for k, v in ta.state_dict().items():
        assert torch.equal(v, tb.state_dict()[k]), f"after duplicate: {k}"
    low = torch.zeros(ta.N, dtype=torch.bool)
    low[::7] = True
    ta.logit_opacities[low.cuda()] = -6.0
    tb.logit_opacities[low.cuda()[tb.ref_index]] = -6.0
    assert ta.cull_opacity(0.05) == tb.cull_opacity(0.05) == int(low.sum())
    for k, v in ta.state_dict().items():
        assert torch.equal(v, tb.state_dict()[k]), f"after cull: {k}"
    ma, mb = ta._moment_views(ta.adam_m), tb._moment_views(tb.adam_m)
    for k in ma:
        assert torch.equal(tb._in_reference_order(mb[k].contiguous()), ma[k].contiguous()), f"moments {k}"
    tb.spatial_sort()                                          # what train() does after an event
    for k, v in ta.state_dict().items():
        assert torch.equal(v, tb.state_dict()[k]), f"after re-sort: {k}"
    ta.ensure_capacity(); tb.ensure_capacity()
    ta.train_step(0, w); tb.train_step(0, w)
    assert abs(ta.pop_loss() - tb.pop_loss()) <= 1e-5 * abs(la)


def test_end_to_end_training_recovers_ground_truth_edges(env):
    """The reference's whole ABC schedule (configs/ABC_DexiNed.json: 400 epochs, every densify / cull /
    regulariser event) on 16 real DexiNed views of scan 00004926 at 400x400, from 2500 random Gaussians:
    the opaque Gaussians must end up ON the ground-truth edge points the reference's eval.py scores
    against (fixture abc_00004926_train.npz; measured precision 0.90-0.92, recall 0.97 at 0.02 over seeds)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_abc_fixture as T
    r = T.run(epochs=400, seed=0)
    precision = float((r["d_pred_to_gt"] < 0.02).float().mean())
    recall = float((r["d_gt_to_pred"] < 0.02).float().mean())
    assert r["steps"] == 6400 and r["n_opaque"] > 1000
    assert r["loss_last"] > 0 and math.isfinite(r["loss_last"])
    assert precision > 0.8 and recall > 0.9, (precision, recall, r["n_final"], r["n_opaque"])


def test_binning_layouts_agree_and_segment_overflow_is_flagged(env):
    """The two binning layouts of the fused step give the same step (same kernels downstream, same sorted
    order per tile); a tile that outgrows its fixed segment drops the excess and raises the flag."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    sc = _scene(synth, n=6000, w=200, h=136, views=2)
    mk = lambda seg: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks,  # noqa: E731
                                 sc.gt, sc.width, sc.height, segmented=seg)
    ta, tb = mk(False), mk(True)
    w = synth.weight_map("weighted", sc.gt[0]).cuda()
    ta.ensure_capacity(); tb.ensure_capacity()
    assert tb.seg_cap > 0 and ta.seg_cap == 0
    ga, gb = ta.grad_step(0, w).clone(), tb.grad_step(0, w).clone()
    assert_close(gb, ga, rtol=1e-5, name="gradients of the two layouts")
    ta.pop_loss(); tb.pop_loss()
    for s in range(3):
        ta.train_step(s % 2, w); tb.train_step(s % 2, w)
    assert ta.last_m() == tb.last_m() and not ta.overflowed() and not tb.overflowed()
    assert int(ta.total[2]) == int(tb.total[2]) and int(ta.total[3]) == int(tb.total[3])
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        # (the two layouts project with different kernels and, since round 3, composite with different ones -- the
        # classic layout keeps the workgroup-per-item forward, the segmented layout runs the wave-autonomous one: same
        # formulas, roundings differ by an ulp, which Adam's epsilon amplifies for the elements whose gradient is ~1e-8:
        # gradients are compared at 1e-5 above, the three-step states at the propagated 1e-4 with that handful admitted)
        assert_close(tb.state_dict()[k], v, rtol=1e-4, max_bad=2e-3, name=k)

    assert int(tb.tile_counts.abs().sum()) == 0, "the segment cursors must be back at zero"
    # overflow: segments far too small for the busiest tiles -> excess dropped, STICKY flag raised, cursors clean
    tb._alloc_isect(tb.capacity, 128)
    assert tb.max_tile_seen > 128
    tb.train_step(0, w)
    tb.train_step(1, w)
    assert tb.overflowed()
    torch.cuda.synchronize()
    assert int(tb.tile_counts.abs().sum()) == 0
    # the read-back notices, grows the segments, restores the state and replays both steps
    ta.train_step(0, w); ta.train_step(1, w)
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert tb.overflow_events >= 1 and not tb.overflowed() and tb.seg_cap > 128
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-4, max_bad=2e-3, name=f"after replay: {k}")  # (as above)


def test_segmented_layout_with_a_giant_tile(env):
    """20000 Gaussians piled onto one spot: a single tile holds them all (> 16384 keys: the hybrid
    global/LDS sort network) inside its fixed segment; the step must equal the scan layout's."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    sc = _scene(synth, n=20000, w=128, h=96, views=1)
    g = torch.Generator().manual_seed(5)
    means = torch.tensor([0.5, 0.5, 0.5]) + 0.004 * torch.randn(20000, 3, generator=g)
    mk = lambda seg: EdgeTrainer(means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks,  # noqa: E731
                                 sc.gt, sc.width, sc.height, segmented=seg)
    ta, tb = mk(False), mk(True)
    ta.ensure_capacity(); tb.ensure_capacity()
    assert tb.seg_cap > 16384 and tb.max_tile_seen > 16384
    w = synth.weight_map("whole", sc.gt[0]).cuda()
    ta.train_step(0, w); tb.train_step(0, w)
    assert not ta.overflowed() and not tb.overflowed() and ta.last_m() == tb.last_m()
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert math.isfinite(la) and abs(la - lb) <= 1e-5 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-5, name=k)


@pytest.mark.parametrize("case", [
    # n, width, height, scale, anisotropy, seed  -- odd image sizes, single Gaussians, fat and thin footprints
    (1, 33, 17, 0.05, 1.0, 1), (7, 16, 16, 0.2, 3.0, 2), (50, 97, 61, 0.08, 8.0, 3), (777, 250, 40, 0.03, 12.0, 4),
    (3000, 61, 203, 0.01, 2.0, 5), (1500, 129, 127, 0.15, 6.0, 6), (4000, 320, 240, 0.004, 5.0, 7),
])
def test_fused_step_vs_operator_on_random_scenes(env, case):
    """The fused step (one-pass segmented binning, slice-parallel forward, footprint backward) against the
    gsplat-compatible operator + torch autograd (count / scan / emit binning, tile-list kernels): two
    disjoint kernel sets on the same inputs, over odd image sizes, 1..4000 Gaussians, footprints from a
    few pixels to most of the image, opacities in (0.05, 0.9) with transmittance stops."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, rasterization
    n, w, h, scale, aniso, seed = case
    sc, _fw, _border, wm, removed = strict_inputs(
        synth.make_scene(n, 2, w, h, seed=seed, spread_opacity=True, scale=scale, anisotropy=aniso), 1, "weighted")
    if removed:  # keep the case's Gaussian count meaningful (n = 1, 7): nothing borderline may be in it
        assert n > 100
    n = sc.means.shape[0]
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, w, h)
    tr.ensure_capacity()
    wm = wm.cuda()
    tr.grad_step(1, wm)
    assert not tr.overflowed()
    got = [t.clone() for t in tr.grad_views()]
    inc = tr.grads.view(-1)[11 * n:].clone()
    P = [sc.means.clone().cuda().requires_grad_(True), sc.quats.clone().cuda().requires_grad_(True),
         sc.log_scales.clone().cuda().requires_grad_(True), sc.logit_opacities.clone().cuda().requires_grad_(True)]
    render, alpha, info = rasterization(P[0], P[1], torch.exp(P[2]), torch.sigmoid(P[3]).squeeze(-1),
                                        torch.ones(n, 3, device="cuda"), sc.viewmats[1:2].cuda(), sc.Ks[1:2].cuda(),
                                        w, h, packed=False, absgrad=True, rasterize_mode="antialiased")
    loss = (wm * (torch.clamp(render[0, ..., 0], 0, 1) - sc.gt[1].cuda()).abs()).sum()
    loss.backward()
    lg = tr.pop_loss()
    assert abs(lg - float(loss)) <= 2e-5 * max(abs(float(loss)), 1e-6), (lg, float(loss))
    ref = (P[0].grad, P[1].grad, P[2].grad, P[3].grad.view(-1))
    floor = 1e-6 * max(float(r.abs().max()) for r in ref)  # a gradient that is zero up to round-off stays "equal"
    for a, b, name in zip(got, ref, ("means", "quats", "scales", "opac")):
        assert torch.isfinite(a).all()
        assert_close(a, b, rtol=1e-4, name=f"{name} {case}", atol_floor=floor)
    assert_close(inc, info["means2d"].absgrad[0].norm(dim=-1), rtol=1e-4, name=f"absgrad {case}")


def _overflow_scene(synth):
    sc = synth.make_scene(6000, 2, 200, 136, seed=3, spread_opacity=False, scale=0.02, anisotropy=5.0)
    return sc


def test_overflow_between_capacity_sweeps_is_replayed(env):
    """Opacities climb from 0.08 to 0.9 between two capacity sweeps (tight tile boxes grow with ln(255 o)): M
    outgrows buffers sized with NO slack.  The sticky device flag is noticed at the read-back, the buffers
    grow, the state is restored and the journalled steps run again: same result as a run that was oversized
    from the start."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = _overflow_scene(synth)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    ta, tb = mk(), mk()
    ta.ensure_capacity(slack=1.0)
    ta._alloc_isect(ta.capacity * 16, ta.seg_cap * 16)       # the oversized run
    tb.ensure_capacity(slack=1.0)                             # sized for opacity 0.08 ...
    tb._alloc_isect(tb.m_max_seen + 64, (tb.max_tile_seen // 128 + 1) * 128)  # ... exactly
    m0 = tb.m_max_seen
    w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(2)]
    tight = (tb.m_max_seen + 64, (tb.max_tile_seen // 128 + 1) * 128)
    for t in (ta, tb):
        t.train_step(0, w[0])                                 # fits
        assert math.isfinite(t.pop_loss()) and t.overflow_events == 0
        t.logit_opacities.fill_(float(torch.logit(torch.tensor(0.9))))   # "training" raises the opacities
    tb._alloc_isect(*tight)                                   # (undo the read-back's look-ahead growth)
    for t in (ta, tb):
        for s in range(4):
            t.train_step(s % 2, w[s % 2])
    assert tb.overflowed() and not ta.overflowed()
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert tb.overflow_events >= 1 and ta.overflow_events == 0 and not tb.overflowed()
    assert tb.last_m() > m0 + 64, "the scene must really have outgrown the first sizing"
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=k)
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-6, name="absgrads")
    assert tb.adam_step == ta.adam_step and tb.step == ta.step and tb.absgrads_normalize_factor == ta.absgrads_normalize_factor


def test_record_table_overflow_of_an_unbalanced_view_is_replayed(env):
    """Round 5, XCD-aware item records: the record table spans 8 x the LONGEST of the eight per-XCD lists.  An object that
    covers a few 2 x 2-tile blocks only puts its slices into a few lists: the span is a multiple of the item count.  The
    count sweep sizes the table for it (`_record_span`); with the table cut back to the item count the sort kernel raises the
    sticky overflow word, the read-back grows the buffers and replays -- same result as the properly sized run."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    W, H = 528, 400  # 33 x 25 = 825 tiles: XCD-aware placement
    sc = synth.make_scene(40_000, 2, W, H, seed=3, anisotropy=3.0, spread_opacity=True, scale=0.006)
    sc.means.mul_(0.15)  # the object shrinks to a few tiles around the image centre (~790 records spanning ~4300 entries)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             W, H, schedule=sched)
    ta, tb = mk(), mk()
    assert _lib.load().eg_record_xcd_shift(ta.T) == 1
    ta.ensure_capacity()
    tb.ensure_capacity()
    tb._rec_need = 0
    tb._alloc_isect(tb.capacity, tb.seg_cap)  # the record table as if every list were as long as the average
    assert tb.max_items < ta._rec_need <= ta.max_items
    w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(2)]
    for t in (ta, tb):
        for s in range(4):
            t.train_step(s % 2, w[s % 2])
    assert tb.overflowed() and not ta.overflowed()
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert tb.overflow_events >= 1 and ta.overflow_events == 0 and not tb.overflowed()
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=k)
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-6, name="absgrads")


def test_rewalk_speculation_miss_is_replayed(env):
    """While no pixel has reached the transmittance stop the trainer does not even launch the exact-stop re-walk
    (EG_REWALK_SPECULATE).  Opacities jump to 0.97 between two read-backs: the first stopping pixel raises the sticky
    miss word, the read-back restores the state and replays the steps with the re-walk on -- same result as a trainer
    that never speculates."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = synth.make_scene(4000, 2, 128, 96, seed=6, spread_opacity=False, scale=0.03, anisotropy=2.0)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda spec: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                                  sc.width, sc.height, schedule=sched, replay_on_overflow=spec)
    ta, tb = mk(False), mk(True)
    ta.ensure_capacity(slack=4.0); tb.ensure_capacity(slack=4.0)
    w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(2)]
    for t in (ta, tb):
        t.train_step(0, w[0])
        assert math.isfinite(t.pop_loss())
    assert tb.rewalk_hint == 0 and tb._rewalk_arg(True) == -2 and ta._rewalk_arg(True) == 0  # tb speculates from now on
    for t in (ta, tb):
        t.logit_opacities.fill_(float(torch.logit(torch.tensor(0.97))))
        t.train_steps([0, 1, 0], [w[0], w[1], w[0]])
        t.train_step(1, w[1])
    assert tb._ctl_bits()[0] and not ta._ctl_bits()[0]
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert getattr(tb, "rewalk_misses", 0) == 1 and tb.rewalk_hint > 0 and not any(tb._ctl_bits())
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=f"after the replay: {k}")
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-6, name="absgrads")
    # ... and the next window runs with the re-walk launched, no further replay
    for t in (ta, tb):
        t.train_step(0, w[0])
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(la) and tb.rewalk_misses == 1


@pytest.mark.parametrize("mode", ["speculative", "chained"])
def test_item_overflow_on_a_large_tile_grid_is_replayed(env, mode):
    """above 2048 tiles the sort kernel writes the item records in two dispatch classes (front slices
    of every tile first); when the view overflows the ITEM table the truncated tiles write no records, and in that order
    the holes would sit below total[2] -- a stale record there sent a wave polling granules nobody publishes (a ~100 ms
    stall and a fatal error instead of grow-and-replay).  The projection's scan now marks an overflowing view
    (item_front[T] = -1) and the sort keeps item order for it; the trainer takes a stall next to an overflow as the
    overflow's.  Both forward modes: same result as a run that was oversized from the start."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    W, H = 1056, 640   # 66 x 40 = 2640 tiles > kPrefixHereMaxTiles (2560 since round 6)
    # (speculative: opacity 0.08 and thin footprints, ~20 layers per pixel -- no pixel reaches the transmittance stop)
    sc = synth.make_scene(40000, 2, W, H, seed=8, anisotropy=5.0, spread_opacity=(mode == "chained"),
                          scale=0.01 if mode == "chained" else 0.004)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    ta, tb = mk(), mk()
    assert ta.T == 2640 and ta.T > _lib.PREFIX_HERE_MAX_TILES
    w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(2)]
    for t in (ta, tb):
        t.ensure_capacity(slack=2.0)
        t.train_step(0, w[0])
        assert math.isfinite(t.pop_loss()) and t.overflow_events == 0
    if mode == "chained":
        assert ta.rewalk_hint != 0, "the trained-like scene must have pixels at the transmittance stop"
    else:
        assert ta.rewalk_hint == 0 and ta._rewalk_arg(True) == -2
    extra = int(ta.total[2]) - ta.T   # items beyond one per tile in the view just rasterised
    assert extra >= 16, extra
    hint = tb.rewalk_hint
    tb._rec_need = 0  # (round 6: the record table of the XCD-aware placement is sized from the count sweep: not here)
    tb._alloc_isect(max(128, extra * 128 * 3 // 4), tb.seg_cap)   # item table: T + 3/4 of what the views need
    tb.rewalk_hint = hint
    assert tb.max_items < int(ta.total[2])
    for t in (ta, tb):
        t.train_steps([1, 0, 1], [w[1], w[0], w[1]])
        t.train_step(0, w[0])
    assert tb.overflowed() and not ta.overflowed()
    la, lb = ta.pop_loss(), tb.pop_loss()    # (before the fix: RuntimeError "a look-back poll gave up")
    assert tb.overflow_events >= 1 and ta.overflow_events == 0 and not tb.overflowed() and not any(tb._ctl_bits())
    assert abs(la - lb) <= 1e-6 * abs(la)
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=f"{mode}: {k}")
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-6, name="absgrads")
    assert tb.adam_step == ta.adam_step and tb.step == ta.step


def test_overflow_without_journal_raises(env):
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    from edgegaussians_amd.trainer import IsectOverflow
    sc = _overflow_scene(synth)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, sc.width, sc.height,
                     replay_on_overflow=False)
    tr.ensure_capacity(slack=1.0)
    tr._alloc_isect(tr.m_max_seen + 64, (tr.max_tile_seen // 128 + 1) * 128)
    w = synth.weight_map("weighted", sc.gt[0]).cuda()
    tr.logit_opacities.fill_(float(torch.logit(torch.tensor(0.9))))
    tr.train_step(0, w)
    tr.logit_opacities.fill_(float(torch.logit(torch.tensor(0.08))))
    tr.train_step(1, w)   # this step fits again: only a STICKY flag still knows about the first one
    with pytest.raises(IsectOverflow):
        tr.pop_loss()
    # the data-parallel leg (grad_step) has no journal either
    tr.logit_opacities.fill_(float(torch.logit(torch.tensor(0.9))))
    tr._alloc_isect(tr.m_max_seen + 64, (tr.max_tile_seen // 128 + 1) * 128)
    tr.grad_step(0, w)
    with pytest.raises(IsectOverflow):
        tr.pop_loss()


def test_train_loop_survives_a_capacity_crossing(env):
    """train() across an opacity ramp with buffers sized without slack: the epoch-end read-back repairs the
    overflow (the loop never looked at the flag)."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, train
    sc = _overflow_scene(synth)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, sc.width, sc.height)
    tr.ensure_capacity(slack=1.0)
    tr._alloc_isect(tr.m_max_seen + 64, (tr.max_tile_seen // 128 + 1) * 128)
    optim = {"means": {"start_lr": 2e-3, "milestones": [], "gamma": 1.0}, "scales": {"start_lr": 1e-4, "start_at_epoch": 0},
             "quats": {"start_lr": 1e-3, "start_at_epoch": 0}, "opacities": {"start_lr": 0.3, "start_at_epoch": 0}}
    training_cfg = {"num_epochs": 3, "optim": optim, "loss": {
        "orientation_losses": {"start_dir_loss_at_epoch": 99, "start_ratio_loss_at_epoch": 99, "dir_loss_num_nn": 5,
                               "dir_loss_scale_factor": 0.01, "ratio_loss_scale_factor": 0.01},
        "projection_losses": {"lambda_annealing": "constant", "lambda_start": 1, "lambda_end": 1,
                              "loss_before_alternating": "whole", "less_freq_loss": "bg_edge_ratio",
                              "more_freq_loss": "whole", "start_alternating_at_epoch": 99,
                              "bg_edge_pixel_ratio_annealing": "constant", "bg_edge_pixel_ratio_start": 1,
                              "bg_edge_pixel_ratio_end": 1, "sampling_whole_num_epochs_ratio": 5}}}
    # a target that wants everything opaque: with lr 0.3 on the logits the opacities run up within an epoch
    tr.gt.fill_(1.0)
    hist = train(tr, {"if_duplicate_high_pos_grad": False, "if_cull_low_opacity": False,
                      "if_cull_gaussians_not_projecting": False}, training_cfg, lambda e: [0, 1] * 20)
    assert len(hist) == 3 and all(math.isfinite(x) for x in hist)
    assert tr.overflow_events >= 1 and not tr.overflowed()
    assert float(torch.sigmoid(tr.logit_opacities).mean()) > 0.5


@pytest.mark.parametrize("Cn,w,h", [(1, 200, 136), (3, 200, 136), (8, 200, 136), (3, 640, 512)])
def test_batched_views_equal_the_sum_of_single_view_steps(env, Cn, w, h):
    """C views in one launch sequence (gridDim.y = view).  Guarantee = the data-parallel one:
    summed gradient == sum of the per-view gradients at the same parameters (same kernels, so to rounding of the
    final sum), loss == sum of losses, absgrad increment == sum of increments; and the batched optimizer step ==
    eg_adam_multi on that sum.  (640 x 512 with 3 views: 3840 tiles in the launch, i.e. the tile scan stays in the
    projection kernel; the small image: the sort kernel forms the prefix.)"""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = _scene(synth, n=4000, w=w, h=h, views=8)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    ta, tb = mk(), mk()
    ta.ensure_capacity(); tb.ensure_capacity()
    views = [5, 0, 3, 7, 1, 2, 6, 4][:Cn]
    strat = ["weighted", "whole", "bg_edge_ratio"]
    wm = [synth.weight_map(strat[i % 3], sc.gt[v], generator=torch.Generator().manual_seed(i)).cuda() for i, v in enumerate(views)]
    N = sc.means.shape[0]
    acc = torch.zeros_like(ta.grads)
    losses = []
    for v, w in zip(views, wm):
        acc += ta.grad_step(v, w)
        losses.append(ta.pop_loss())
    gb = tb.grad_step_batched(views, wm).clone()
    lb = tb.pop_loss()
    assert abs(lb - sum(losses)) <= 1e-5 * abs(sum(losses)) and not tb.overflowed()
    for name, sl in (("means", slice(0, 3 * N)), ("quats", slice(3 * N, 7 * N)), ("scales", slice(7 * N, 10 * N)),
                     ("opac", slice(10 * N, 11 * N)), ("absgrad", slice(11 * N, 12 * N))):
        assert_close(gb.view(-1)[sl], acc.view(-1)[sl], rtol=1e-5, name=f"batched {name}")
    # the batched optimizer step == Adam on the summed gradient
    ta.grads.copy_(acc)
    ta.apply_adam()
    tb.train_step_batched(views, wm)
    assert tb.adam_step == ta.adam_step == 1 and tb.step == 2 * Cn
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=f"batched step {k}")
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-5, name="batched absgrads")
    assert tb.absgrads_normalize_factor == 1 + Cn
    assert math.isfinite(tb.pop_loss())
    assert int(tb._batches[Cn]["tile_counts"].abs().sum()) == 0 and int(tb._batches[Cn]["ticket"].abs().sum()) == 0


def test_device_weight_maps_match_the_host_construction(env):
    """EdgeTrainer.weight_map on the device (eg_ratio_wmap for 'bg_edge_ratio') against synth.weight_map (the host
    construction pinned to the reference by tests/test_golden.py): same values for the deterministic strategies; for
    the sampled one the edge part is identical and the sample is n_sel DISTINCT pixels of weight 1/n_sel."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    sc = _scene(synth, n=500, w=200, h=136, views=2)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, sc.width, sc.height)
    for strat in ("whole", "weighted"):
        assert_close(tr.weight_map(1, strat), synth.weight_map(strat, sc.gt[1]), rtol=1e-6, name=strat)
    edge = sc.gt[1] >= 0.5
    n_e, hw = int(edge.sum()), edge.numel()
    for ratio in (1.0, 0.5, 3.0):
        w = tr.weight_map(1, "bg_edge_ratio", ratio).cpu()
        n_sel = int(ratio * n_e)
        sel_part = w - edge.float() / n_e
        picked = sel_part > 0.5 / n_sel
        assert int(picked.sum()) == n_sel, (int(picked.sum()), n_sel)
        assert_close(sel_part[picked], torch.full((n_sel,), 1.0 / n_sel), rtol=1e-5, name="sample weight")
        assert float(sel_part[~picked].abs().max()) < 1e-9
        assert abs(float(w.sum()) - 2.0) < 1e-4
        # the reference's quirk (edge_gs.py:303-310): indices are drawn from [0, #bg) and unravelled over the whole image
        assert int(torch.nonzero(picked.view(-1)).max()) < hw - n_e
    a, b = tr.weight_map(1, "bg_edge_ratio", 1.0), tr.weight_map(1, "bg_edge_ratio", 1.0)
    assert not torch.equal(a, b), "every call draws a fresh sample"
    # the sample is uniform over [0, #bg): over D draws every pixel is hit ~ D n_sel / n_bg times (binomial)
    D, n_bg = 400, hw - n_e
    hits = torch.zeros(hw)
    for _ in range(D):
        hits += (tr.weight_map(1, "bg_edge_ratio", 1.0).cpu().view(-1) - edge.float().view(-1) / n_e > 0.5 / n_e).float()
    p = n_e / n_bg
    mean, sd = D * p, math.sqrt(D * p * (1 - p))
    z = (hits[:n_bg] - mean) / sd
    assert float(hits[n_bg:].sum()) == 0
    assert abs(float(z.mean())) < 0.05 and 0.9 < float(z.std()) < 1.1, (float(z.mean()), float(z.std()))
    assert float(z.abs().max()) < 6.0
    # ... and consecutive draws are independent: the overlap of two samples is ~ n_sel^2 / n_bg
    s1 = tr.weight_map(1, "bg_edge_ratio", 1.0).cpu().view(-1) - edge.float().view(-1) / n_e > 0.5 / n_e
    s2 = tr.weight_map(1, "bg_edge_ratio", 1.0).cpu().view(-1) - edge.float().view(-1) / n_e > 0.5 / n_e
    both, expect = int((s1 & s2).sum()), n_e * n_e / n_bg
    assert abs(both - expect) < 6 * math.sqrt(expect) + 3, (both, expect)
    # round 6: a run's maps by ONE native call (weight_maps -> eg_ratio_wmaps_seeded) = the same maps, the same draw sequence
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, sc.width, sc.height)  # noqa: E731
    t1, t2 = mk(), mk()
    vs, st = [1, 0, 0, 1, 1, 0], ["bg_edge_ratio", "whole", "bg_edge_ratio", "weighted", "bg_edge_ratio", "bg_edge_ratio"]
    one_by_one = [t1.weight_map(v, k, 0.7) for v, k in zip(vs, st)]
    batch = t2.weight_maps(vs, st, 0.7)
    for x, y in zip(one_by_one, batch):
        assert y.is_contiguous() and y.shape == (sc.height, sc.width) and torch.equal(x, y)
    assert t1._wmap_draws == t2._wmap_draws == 4 and torch.equal(t1.weight_map(0, "bg_edge_ratio"), t2.weight_maps([0], ["bg_edge_ratio"])[0])


def test_native_run_of_steps_equals_single_steps(env):
    """EdgeTrainer.train_steps (eg_train_steps: K reference iterations per native call, each step's last kernel
    also projecting + binning the next view) against K calls of train_step: same kernels' arithmetic, same state."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = _scene(synth, n=4000, w=200, h=136, views=5)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    ta, tb = mk(), mk()
    ta.ensure_capacity(); tb.ensure_capacity()
    views = [3, 0, 4, 1, 2, 0, 3]
    wm = [synth.weight_map(("weighted", "whole", "bg_edge_ratio")[i % 3], sc.gt[v], generator=torch.Generator().manual_seed(i)).cuda()
          for i, v in enumerate(views)]
    for v, w in zip(views, wm):
        ta.train_step(v, w)
    tb.train_steps(views[:1], wm[:1])       # K = 1: no tail
    tb.train_steps(views[1:], wm[1:])       # K = 6: five fused tails
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(la) and not tb.overflowed()
    assert tb.adam_step == ta.adam_step == 7 and tb.step == 7 and tb.group_steps == ta.group_steps
    assert tb.absgrads_normalize_factor == ta.absgrads_normalize_factor
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=f"run of steps: {k}")
    assert_close(tb.absgrads, ta.absgrads, rtol=1e-6, name="absgrads")
    assert_close(tb.adam_v, ta.adam_v, rtol=1e-6, name="second moments")
    # a following single step starts from scratch (have_projection = 0): still the same state
    ta.train_step(1, wm[0]); tb.train_step(1, wm[0])
    for k, v in ta.state_dict().items():
        assert_close(tb.state_dict()[k], v, rtol=1e-6, name=f"after the run: {k}")
    assert int(tb.tile_counts.abs().sum()) == 0 and int(tb.ticket[0]) == 0


@pytest.mark.parametrize("case", ["small", "stops", "config1_size", "tiny", "wide_footprints", "native_800"])
def test_fused_backward_kernel_equals_the_two_kernel_path(env, case):
    """Round 6: inside a native run of steps the backward of a scene of <= 32768 Gaussians is ONE kernel (csrc/backward_fused.hip:
    footprint backward, then -- in the workgroup's first wave -- projection backward + absgrads + Adam + the next view's
    projection and binning).  It inlines the functions the two kernels of rounds 1-5 inline: every parameter, moment and
    absgrad must come out BIT FOR BIT the same as with eg_step_args.two_kernel_backward, the loss sums to the order of their
    float atomics."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    if case == "small":
        sc = _scene(synth, n=4001, w=200, h=136, views=5)          # (N not a multiple of 64: a ragged last workgroup)
    elif case == "stops":
        sc = synth.make_scene(3000, 4, 128, 96, seed=6, spread_opacity=False, scale=0.03, anisotropy=2.0)
        sc.logit_opacities[:] = torch.logit(torch.tensor(0.97))    # transmittance stops: the order-dependent part
    elif case == "tiny":
        sc = synth.make_scene(37, 4, 33, 17, seed=3, scale=0.05)    # one partial workgroup, 3 x 2 tiles
    elif case == "native_800":
        sc = synth.make_scene(9000, 4, 800, 800, seed=12, spread_opacity=True)   # the reference's native size: 2500 tiles
    elif case == "wide_footprints":
        # 100 Gaussians that each cover most of a 45 x 25-tile grid: a workgroup's 64 Gaussians touch more tiles than the
        # touched list holds (512) -- the wave sweeps the whole histogram instead
        sc = synth.make_scene(100, 4, 720, 400, seed=4, scale=0.25, anisotropy=1.5, spread_opacity=True)
    else:
        sc = synth.make_scene(30011, 6, 512, 512, seed=11)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched, spatial_order=(case == "config1_size"))
    ta, tb = mk(), mk()
    ta.two_kernel_backward = 1   # rounds 1-5: footprint backward, then projection backward with 512 Gaussians per workgroup
    assert not tb.two_kernel_backward and tb.N <= 32768 and tb.T <= 2560
    ta.ensure_capacity(); tb.ensure_capacity()
    assert tb.fused_backward_active() and not ta.fused_backward_active()
    V = sc.viewmats.shape[0]
    views = [(3 * i + 1) % V for i in range(9)]
    wm = [synth.weight_map(("weighted", "whole", "bg_edge_ratio")[i % 3], sc.gt[v], generator=torch.Generator().manual_seed(i)).cuda()
          for i, v in enumerate(views)]
    for t in (ta, tb):
        t.train_steps(views[:4], wm[:4])
        t.train_steps(views[4:], wm[4:])
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(la), (la, lb)  # (the loss terms meet in float atomics: the sum's last bits follow their order)
    assert not ta.overflowed() and not tb.overflowed()
    sa, sb = ta.state_dict(), tb.state_dict()
    for k, v in sa.items():
        assert torch.equal(sb[k], v), f"fused backward kernel: {k} differs from the two-kernel path"
    for name in ("absgrads", "adam_m", "adam_v"):
        assert torch.equal(getattr(tb, name), getattr(ta, name)), name
    # a single step after the run (no next view: the footprint backward + project_bwd_adam) leaves both in the same state again
    ta.train_step(views[0], wm[0]); tb.train_step(views[0], wm[0])
    for k, v in ta.state_dict().items():
        assert torch.equal(tb.state_dict()[k], v), f"after the run: {k}"
    assert int(tb.tile_counts.abs().sum()) == 0
    if case == "wide_footprints":
        assert tb.last_m() > 512 * 2, "the scene must touch more tiles per workgroup than the touched list holds"


def test_batched_views_with_stops_and_overflow_replay(env):
    """The batched launch sequence on a stop-heavy scene (exact-stop re-walk per view, gridDim.y) equals the sum of the
    single-view steps; and a batched step that overflows buffers sized without slack is replayed from the journal."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = synth.make_scene(4000, 4, 128, 96, seed=6, spread_opacity=False, scale=0.03, anisotropy=2.0)
    sc.logit_opacities[:] = torch.logit(torch.tensor(0.97))
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched)
    ta, tb = mk(), mk()
    ta.ensure_capacity(); tb.ensure_capacity()
    views = [2, 0, 3]
    wm = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in views]
    acc = torch.zeros_like(ta.grads)
    for v, w in zip(views, wm):
        acc += ta.grad_step(v, w)
    la = ta.pop_loss()
    gb = tb.grad_step_batched(views, wm).clone()
    lb = tb.pop_loss()
    assert abs(la - lb) <= 1e-5 * abs(la)
    assert_close(gb, acc, rtol=1e-5, name="batched gradients on a stop-heavy scene")
    assert max(b["rewalk_hint"] for b in tb._batches.values()) > 0, "the scene must exercise the re-walk"
    # overflow inside a batched step: buffers sized for opacity 0.08, opacities jump to 0.9
    sc2 = synth.make_scene(6000, 3, 200, 136, seed=3, spread_opacity=False, scale=0.02, anisotropy=5.0)
    mk2 = lambda: EdgeTrainer(sc2.means, sc2.log_scales, sc2.quats, sc2.logit_opacities, sc2.viewmats, sc2.Ks, sc2.gt,  # noqa: E731
                              sc2.width, sc2.height, schedule=sched)
    t1, t2 = mk2(), mk2()
    t1.ensure_capacity(slack=1.0)
    t1._alloc_isect(t1.capacity * 16, t1.seg_cap * 16)
    t2.ensure_capacity(slack=1.0)
    t2._alloc_isect(t2.m_max_seen + 64, (t2.max_tile_seen // 128 + 1) * 128)
    w2 = [synth.weight_map("weighted", sc2.gt[v]).cuda() for v in range(3)]
    for t in (t1, t2):
        t.logit_opacities.fill_(float(torch.logit(torch.tensor(0.9))))
        t.train_step_batched([0, 1, 2], w2)
        t.train_step_batched([2, 0], w2[:2])
    assert t2.overflowed() and not t1.overflowed()
    l1, l2 = t1.pop_loss(), t2.pop_loss()
    assert t2.overflow_events >= 1 and t1.overflow_events == 0 and not t2.overflowed()
    assert abs(l1 - l2) <= 1e-6 * abs(l1) and t2.adam_step == t1.adam_step == 2
    for k, v in t1.state_dict().items():
        assert_close(t2.state_dict()[k], v, rtol=1e-5, name=f"batched replay: {k}")


def test_bench_line_contract(env):
    """bench.py prints ONE JSON line, last on stdout, with the fields the driver and the judge read."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "config1", "--steps", "20",
                        "--warmup", "5", "--cpu-budget", "1", "--no-traffic", "--no-extra"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["n_gaussians"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert "untuned" in cb["sample"].lower()  # (a stated baseline, not a speed-up denominator)
    # round 3: what happened before and around the timed window is on the line
    assert d["prewarm_steps"] >= 400 and len(d["ms_per_step_windows"]) == 3
    assert d["rehearsal_steps"] == 5 + 200  # (round 6: an untimed window of the timed one's shape runs in front of it)
    assert d["ms_per_step_windows"][0] == d["ms_per_step"]  # `value` is the contract's window, the repeats are extra
    assert d["ms_per_step_min"] == min(d["ms_per_step_windows"]) and d["ms_per_step_median"] in d["ms_per_step_windows"]
    assert d["config"]["forward_mode"].startswith(("speculative", "chained")) and "opacity" in d["config"]["workload"]
    # round 4: a window of fewer than 200 steps says so on the line
    assert "warning" in d and "--steps 20" in d["warning"]
    # round 5: M on the line is a MEAN over the views a measurement covered (the last view's M moved frac by 10 % from run
    # to run), and the roofline's algorithmic bytes follow from the M they name
    cfg = d["config"]
    lo, hi = cfg["tile_intersections_M_min_max"]
    assert lo <= cfg["tile_intersections_M"] <= hi and lo <= cfg["tile_intersections_M_all_views"] <= hi and lo < hi
    assert lo <= rf["algorithmic_bytes_M"] <= hi
    if rf["kernel"] == "composite_slice_fwd":
        assert abs(rf["algorithmic_bytes_per_launch"] - (28 * rf["algorithmic_bytes_M"] + 20 * cfg["width"] * cfg["height"])) < 1.0
    if rf["kernel"] == "footprint_bwd+project_bwd_adam+next_project_bin":
        # round 6, the one-kernel backward: G8's gather + per-pixel records, G9 + absgrad + Adam, the next view's G1 + keys; the
        # g2d record's 64 N bytes (written by one kernel, read by the next until round 5) are gone
        assert rf["kernel_symbol"] == "gaussian_bwd_fused_kernel"
        assert abs(rf["algorithmic_bytes_per_launch"] - (562 * cfg["n_gaussians"] + 40 * rf["algorithmic_bytes_M"] + 20 * cfg["width"] * cfg["height"])) < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * rf["achieved"]


def test_bench_line_carries_the_target_configuration(env, golden_dir):
    """Round 6 (VERDICT r05 missing 3 / 4): north_star's named configuration -- config 1, ~30 k Gaussians on the scan's 50 views
    @512 x 512 -- rides on the line as a full citizen (value, its own roofline, its own CPU baseline on the same N), and the scan
    at its NATIVE 800 x 800 against its real DexiNed maps is a workload of the same line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "200", "--warmup", "10", "--cpu-budget", "2",
                        "--no-traffic", "--extra-set", "target"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["n_gaussians"] == 100_000  # the headline stays config 2
    for k in ("value_config1", "ms_per_step_config1", "roofline_config1", "cpu_baseline_config1", "cpu_baseline"):
        assert k in d, k
    c1 = d["other_workloads"]["config1"]
    assert c1["config"]["n_gaussians"] == 30_000 and abs(d["value_config1"] - 30_000 / (d["ms_per_step_config1"] * 1e-3)) < 1e-6 * d["value_config1"]
    cb1 = d["cpu_baseline_config1"]
    assert cb1["kind"] == "port" and cb1["value"] > 0 and "N=30000" in cb1["sample"] and cb1["unit"] == d["unit"]
    rf1 = d["roofline_config1"]
    assert rf1["bound"] == "hbm" and 0 < rf1["frac"] < 1 and "traffic" in rf1
    a8 = d["other_workloads"]["abc800_real_edges"]
    assert a8["config"]["width"] == 800 and a8["config"]["height"] == 800 and a8["value"] > 0
    assert "REAL DexiNed" in a8["config"]["workload"] and "as recorded" in a8["config"]["workload"]
    assert a8["config"]["tile_intersections_M"] > 0 and len(a8["ms_per_step_windows"]) == 3


def test_step_on_a_cu_masked_device_is_identical(env, tmp_path):
    """Round 6: the forward's look-back relies on in-order workgroup dispatch (a slice only waits for
    lower-indexed workgroups) and places its records by `workgroup b -> XCD b % 8`; no masked or partitioned device had ever run
    it.  tools/cu_mask_check.py trains 24 steps of a stop-heavy 20 k-Gaussian scene (chained forward) in a fresh process: once
    plain, once on 32 CUs (ROC_GLOBAL_CU_MASK: the runtime then reports 32) and once under a scattered HSA_CU_MASK -- the
    parameters must come out identical, the bounded look-back poll must not have given up, nothing replayed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("plain", {}), ("global32", {"ROC_GLOBAL_CU_MASK": "0xffffffff"}),
                       ("scattered", {"HSA_CU_MASK": "0:0-7,32-39,64-71,200-207"})):
        envv = {k: v for k, v in os.environ.items() if k not in ("ROC_GLOBAL_CU_MASK", "HSA_CU_MASK")}
        envv.update(extra)
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "cu_mask_check.py"), path], capture_output=True, text=True,
                           timeout=300, cwd=root, env=envv)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (np.load(path), r.stdout.strip().splitlines()[-1])
    assert "CUs reported 32;" in outs["global32"][1], outs["global32"][1]
    a = outs["plain"][0]
    assert int(a["rewalk_hint"]) != 0, "the scene must have pixels at the transmittance stop (chained forward)"
    for tag in ("global32", "scattered"):
        b = outs[tag][0]
        assert int(b["stall"]) == 0, f"{tag}: a look-back poll gave up"
        for k in a.files:
            if k not in ("loss",):
                assert np.array_equal(a[k], b[k]), f"{tag}: {k} differs from the unmasked run"
        assert abs(float(a["loss"]) - float(b["loss"])) <= 1e-6 * abs(float(a["loss"]))
        assert "replays 0" in outs[tag][1]


def test_bench_launches_its_own_ranks(env):
    """`python bench.py --gpus 2` WITHOUT a launcher (WORLD_SIZE unset) starts its own two ranks under
    torch.distributed.run and prints a line for TWO ranks (round 5, VERDICT r04 item 1: it used to run one rank and say
    n_gpus 1).  On this one-GPU box the ranks share the device over gloo (EG_DIST_BACKEND: RCCL refuses that), which
    runs the Python driver; the native RCCL leg needs one device per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envv = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    envv["EG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "config1", "--steps", "20",
                        "--warmup", "5", "--no-traffic", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=envv)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, "one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["views_per_step"] == 2 and d["steps"] == 20
    assert d["config"]["parallelism"].startswith("dp2") and d["config"]["data_parallel_leg"].startswith("python")
    pr = d["collective_proof"]
    assert pr["torch_distributed_world_size"] == 2 and pr["torch_distributed_backend"] == "gloo"
    assert sorted(x[0] for x in pr["rank_device_pid"]) == [0, 1] and len({x[2] for x in pr["rank_device_pid"]}) == 2
    assert len(d["allreduce_exposed_us_per_step_by_rank"]) == 2 and d["allreduce_bytes_per_step"] == 48 * d["config"]["n_gaussians"]
    assert abs(d["value"] - 2 * d["config"]["n_gaussians"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # round 6: whatever the first scaling run shows, the line says what the collective cost -- the same enqueue path timed
    # with the gradient all-reduce left out on every rank (run last: the replicas diverge in it) -- and carries the
    # overlapped two-half-batches mode as a sibling
    nc = d["no_collective_leg"]
    assert d["ms_per_step_no_collective"] > 0 and nc["steps"] >= 20 and "MEASUREMENT ONLY" in nc["what"]
    assert abs(nc["non_collective_share_of_step"] - d["ms_per_step_no_collective"] / d["ms_per_step_median"]) < 1e-9
    assert abs(nc["collective_cost_us_per_step"] - 1e3 * (d["ms_per_step_median"] - d["ms_per_step_no_collective"])) < 1e-6
    v2 = d["views_per_step_2"]
    assert v2["config"]["views_per_step"] == 4 and v2["value"] > 0 and "ms_per_step_no_collective" in v2
    assert abs(v2["value"] - 4 * v2["config"]["n_gaussians"] / (v2["ms_per_step"] * 1e-3)) <= 1e-6 * v2["value"]
    # ... and without the test mode it REFUSES: two ranks asked for, one device visible
    envv.pop("EG_DIST_BACKEND")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--steps", "5"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=envv)
    assert r.returncode != 0 and "device(s) are visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_regulariser_step_with_order_independent_sums(env):
    """eg_regulariser_step_fixed (round 4): the direction loss's neighbour gradients and both loss sums accumulated with
    64-bit fixed-point integer atomics.  (1) Bit-identical from run to run -- what a data-parallel run needs of its
    replicas, where float atomics differ in the last bits with the order the hardware takes them; (2) equal to the float
    path to ~1e-6 after a step (the two differ by the quantisation, 2^-32 per term, and by the float path's own order)."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    sc = _scene(synth, n=6000, w=96, h=80)
    sc.log_scales[:, 0] += 1.0  # (a clear major axis)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)

    def run(det, kinds=("direction", "ratio", "direction")):
        tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, 96, 80, schedule=sched)
        tr.deterministic_regularisers = det
        tr.epoch = 60
        w = synth.weight_map("weighted", sc.gt[0]).cuda()
        tr.train_step(0, w)
        # (the regulariser's weight is formed on the device from the running projection-loss sum, and THAT sum is not
        # bit-reproducible: the forward's waves meet in 64 partial sums through float atomics, three or more to a slot -- one
        # run in twenty differed in its last bit in round 5's tree already, nearly every run once round 6's smaller projection
        # workgroups changed the timing.  What this test pins is the regulariser's own sums: same loss sum in, same bits out.)
        tr.loss_acc.fill_(0.0078125)
        vals = [tr.regulariser_step(k, None, 0.01, 5, "enforce_full") for k in kinds]
        torch.cuda.synchronize()
        return tr, vals

    a, va = run(True)
    b, vb = run(True)
    for x, y in ((a.means, b.means), (a.quats, b.quats), (a.log_scales, b.log_scales), (a.adam_m, b.adam_m), (a.adam_v, b.adam_v)):
        assert torch.equal(x, y)
    assert va == vb
    c, vc = run(False)
    for x, y, name in ((a.means, c.means, "means"), (a.quats, c.quats, "quats"), (a.log_scales, c.log_scales, "scales")):
        assert_close(x, y, rtol=2e-6, name=name)
    for p_, q_ in zip(va, vc):
        assert abs(p_ - q_) <= 1e-5 * abs(q_)
    assert int(a._reg_fixed.abs().sum()) == 0, "the fixed-point scratch is handed back zeroed"


def test_operator_deferred_read_back_is_exact_and_fails_loudly(env):
    """The fast path's verdicts (overflow, colours == 1) are read back asynchronously once two consecutive calls came back
    clean with room to spare (rasterizer.py: DEFERRED READ-BACK).  (1) Calls in deferred mode return exactly what calls that
    synchronise return.  (2) A verdict that turns out bad after the outputs were handed out RAISES at the next look (the
    backward, or the next call) -- never silently wrong."""
    _lib, synth, O = env
    from edgegaussians_amd import rasterization
    from edgegaussians_amd import rasterizer as R
    sc = _scene(synth, n=2000, w=112, h=80, views=3)
    dev = "cuda"
    p = [t.clone().to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)

    def run(view, colors):
        for t in p:
            t.grad = None
        render, alpha, info = rasterization(means=p[0], quats=p[1], scales=torch.exp(p[2]), opacities=torch.sigmoid(p[3]).squeeze(-1),
                                            colors=colors, viewmats=vm[view:view + 1], Ks=K[view:view + 1], width=112, height=80,
                                            packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        (render[0, ..., 0] * sc.gt[view].to(dev)).sum().backward()
        return render.detach().clone(), [t.grad.clone() for t in p], info["means2d"].absgrad.clone()

    ones = lambda: torch.ones(2000, 3, device=dev)  # noqa: E731
    R._FAST.clear()
    outs = [run(v % 3, ones()) for v in range(6)]
    fb = next(iter(R._FAST.values()))
    assert fb.confident >= 2 and not fb.pending, "calls 3.. must have run with a deferred read-back (settled in backward)"
    old = R._DEFER
    try:
        R._DEFER = False
        R._FAST.clear()
        ref = [run(v % 3, ones()) for v in range(6)]
    finally:
        R._DEFER = old
    for a, b in zip(outs, ref):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    # (2) colours that stop being ones while the path is in deferred mode
    R._FAST.clear()
    for v in range(4):
        run(v % 3, ones())
    bad = ones()
    bad[5, 1] = 0.5
    with pytest.raises(RuntimeError, match="not all ones"):
        run(0, bad)       # raised by the backward's look at the forward's verdicts
    out = run(1, bad)     # the path has fallen back to reading back at once: the general operator takes the call
    assert float((out[0][0, ..., 0] - out[0][0, ..., 1]).abs().max()) > 0  # (colours really differ per channel)
    # (3) round 5: a call nobody will run a backward through -- grad mode off: the evaluation renders after
    # training -- never defers: its verdicts are read before it returns, so real colours get the general operator at once
    R._FAST.clear()
    for v in range(4):
        run(v % 3, ones())
    fb = next(iter(R._FAST.values()))
    assert fb.confident >= 2
    with torch.no_grad():
        render, _alpha, _info = rasterization(means=p[0], quats=p[1], scales=torch.exp(p[2]), opacities=torch.sigmoid(p[3]).squeeze(-1),
                                              colors=bad, viewmats=vm[1:2], Ks=K[1:2], width=112, height=80, packed=False,
                                              absgrad=True, rasterize_mode="antialiased")
    assert not fb.pending and torch.equal(render, out[0])
    # (4) ... and a forward whose backward never runs is looked at when a call for ANOTHER shape arrives (N changed)
    for v in range(3):
        run(v % 3, ones())
    render, _a, _i = rasterization(means=p[0], quats=p[1], scales=torch.exp(p[2]), opacities=torch.sigmoid(p[3]).squeeze(-1),
                                   colors=bad, viewmats=vm[0:1], Ks=K[0:1], width=112, height=80, packed=False, absgrad=True,
                                   rasterize_mode="antialiased")   # deferred, wrong, and no backward follows
    assert fb.pending
    q = [t.detach()[:1500].clone().requires_grad_(True) for t in p]
    with pytest.raises(RuntimeError, match="not all ones"):
        rasterization(means=q[0], quats=q[1], scales=torch.exp(q[2]), opacities=torch.sigmoid(q[3]).squeeze(-1),
                      colors=torch.ones(1500, 3, device=dev), viewmats=vm[0:1], Ks=K[0:1], width=112, height=80, packed=False,
                      absgrad=True, rasterize_mode="antialiased")
    R._FAST.clear()


def test_scenes_side_by_side_on_streams_and_threads_equal_their_solo_runs(env):
    """BASELINE config 5 on one GPU (bench.py --scenes-per-gpu): S independent EdgeTrainers, each on its own HIP stream and
    driven by its own host thread, the S launch sequences running concurrently.  Nothing is shared: every trainer ends
    bit-identical to the same trainer run alone on the default stream."""
    import threading
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    S, K = 3, 24
    scs = [_scene(synth, n=2500 + 300 * i, w=160, h=112, views=3, seed=10 + i) for i in range(S)]
    mk = lambda sc: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                                sc.width, sc.height)
    views = [k % 3 for k in range(K)]

    def steps(tr, sc):
        w = [synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(3)]
        torch.cuda.current_stream().synchronize()
        for k0 in range(0, K, 8):
            tr.train_steps(views[k0:k0 + 8], [w[v] for v in views[k0:k0 + 8]])
        return tr.pop_loss()

    solo = []
    for sc in scs:
        tr = mk(sc)
        solo.append((steps(tr, sc), tr))
    side, streams, out = [mk(sc) for sc in scs], [torch.cuda.Stream() for _ in scs], [None] * S

    def drive(i):
        with torch.cuda.stream(streams[i]):
            out[i] = steps(side[i], scs[i])

    th = [threading.Thread(target=drive, args=(i,)) for i in range(S)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    for i in range(S):
        a, b = solo[i][1], side[i]
        assert abs(out[i] - solo[i][0]) <= 1e-6 * abs(solo[i][0]), (i, out[i], solo[i][0])  # (a sum of float atomics)
        for x, y in ((a.means, b.means), (a.log_scales, b.log_scales), (a.quats, b.quats), (a.logit_opacities, b.logit_opacities),
                     (a.adam_m, b.adam_m), (a.adam_v, b.adam_v), (a.absgrads, b.absgrads)):
            assert torch.equal(x, y), i


@pytest.mark.parametrize("n_threads", [1, 2, 0])
def test_native_multi_scene_run_equals_the_solo_runs(env, n_threads):
    """eg_train_steps_multi (BASELINE config 5 with S scenes per GPU, one native call for K steps of every scene, round-robin
    over S streams from 1 / 2 / S host threads inside the call): every trainer ends bit-identical to its solo run; an
    argument error of one scene names the scene."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer, train_steps_multi
    S, K = 3, 24
    scs = [_scene(synth, n=2500 + 300 * i, w=160, h=112, views=3, seed=10 + i) for i in range(S)]
    mk = lambda sc: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                                sc.width, sc.height)
    views = [k % 3 for k in range(K)]
    wm = [[synth.weight_map("weighted", sc.gt[v]).cuda() for v in range(3)] for sc in scs]
    solo = []
    for i, sc in enumerate(scs):
        tr = mk(sc)
        for k0 in range(0, K, 8):
            tr.train_steps(views[k0:k0 + 8], [wm[i][v] for v in views[k0:k0 + 8]])
        solo.append((tr.pop_loss(), tr))
    side, streams = [mk(sc) for sc in scs], [torch.cuda.Stream() for _ in scs]
    torch.cuda.synchronize()
    for k0 in range(0, K, 8):
        train_steps_multi(side, [views[k0:k0 + 8]] * S, [[wm[i][v] for v in views[k0:k0 + 8]] for i in range(S)], streams,
                          n_threads=n_threads)
    torch.cuda.synchronize()
    for i in range(S):
        a, b = solo[i][1], side[i]
        with torch.cuda.stream(streams[i]):
            loss = b.pop_loss()
        assert abs(loss - solo[i][0]) <= 1e-6 * abs(solo[i][0]), (i, loss, solo[i][0])  # (a sum of float atomics)
        for x, y in ((a.means, b.means), (a.log_scales, b.log_scales), (a.quats, b.quats), (a.logit_opacities, b.logit_opacities),
                     (a.adam_m, b.adam_m), (a.adam_v, b.adam_v), (a.absgrads, b.absgrads)):
            assert torch.equal(x, y), i
    with pytest.raises(RuntimeError, match="scene 1"):  # (view index -1 of scene 1: the native call names it)
        train_steps_multi(side, [[0], [-1], [0]], [[wm[i][0]] for i in range(S)], streams, n_threads=n_threads)
    torch.cuda.synchronize()


def test_roctx_ranges_do_not_change_a_step(env):
    """eg_roctx_enable(1) wraps the stages of eg_train_step in roctx ranges (tracing; rocprofv3 --marker-trace);
    with or without them the step is the same."""
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    sc = _scene(synth, n=1500, w=96, h=80)
    w = synth.weight_map("weighted", sc.gt[0]).cuda()
    out = []
    for on in (0, 1):
        assert _lib.load().eg_roctx_enable(on) == 0, _lib.load().eg_last_error_string()
        tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, 96, 80)
        tr.train_step(0, w)
        tr.train_steps([1, 0], [w, w])
        out.append((tr.means.clone(), tr.adam_v.clone(), tr.pop_loss()))
    _lib.load().eg_roctx_enable(0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert abs(out[0][2] - out[1][2]) <= 1e-6 * abs(out[0][2])


def test_bench_kernel_trace_pass(env):
    """`roofline.avg_launch_us` comes from a same-run rocprofv3 --kernel-trace pass over bench.py itself (the figure the
    committed profiles hold; HIP events read a few us more per pair): the pass returns every kernel of the step."""
    import shutil
    import bench
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    kt, src = bench.measure_kernel_trace("config1", False, steps=40)
    assert kt is not None, src
    # (config 1 -- 30 k Gaussians -- runs its backward as ONE kernel inside a native run, round 6; the run's last step has no
    # next view to project and takes the two kernels)
    for k in ("composite_wave_fwd_kernel", "gaussian_bwd_fused_kernel", "footprint_bwd_kernel", "tile_sort_kernel"):
        assert k in kt and 1.0 < kt[k] < 200.0, (k, kt)
    assert "kernel-trace" in src


def test_bench_issue_roofline_pass(env):
    """`roofline.secondary` (the issue-side roofline, from a same-run rocprofv3 SQ-counter pass over bench.py itself):
    every kernel of the step with its VALU wave-instructions per launch, its fraction of the chip's issue peak and
    the split of its wave cycles."""
    import shutil
    import bench
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    kernels, src = bench.measure_issue("config1", False, steps=10)
    assert kernels is not None, src
    assert "composite_wave_fwd_kernel" in kernels and "gaussian_bwd_fused_kernel" in kernels and "tile_sort_kernel" in kernels
    for k, r in kernels.items():
        assert r["valu_wave_instructions_per_launch"] > 0 and 0 < r["valu_issue_frac_of_peak"] < 1, (k, r)
        shares = r["wave_cycles_issuing"] + r["wave_cycles_parked_waitcnt_or_barrier"] + r["wave_cycles_issue_stalled"]
        assert 0.9 < shares < 1.1, (k, r)
    assert bench.VALU_ISSUE_PEAK == 256 * 4 * 2.4e9 / 2.0


@pytest.mark.parametrize("W,H,n", [(1024, 512, 8000), (1040, 512, 8000), (1280, 512, 8000), (1296, 512, 8000), (800, 800, 8000),
                                   (2048, 16, 3000), (33, 17, 500), (721, 403, 6000)])
def test_tile_grid_shapes_around_the_prefix_switch(env, W, H, n):
    """Tile grids on either side of kPrefixHereMaxTiles (2560 tiles since round 6 -- the prefix in one, two or three batches of
    1024 cursors; 800 x 800 is the reference's native size: the sort kernel forms the tile prefix and the
    compositing kernel returns the cursors to zero; above it the projection kernel's last workgroup scans), strips one
    tile high / wide, and sizes that are not multiples of 16: one fused step against the plain-C oracle, then a
    native run of steps (tail fusion across the step boundary) that must leave every cursor at zero."""
    _lib, synth, O = env
    sc = synth.make_scene(n, 2, W, H, seed=5, anisotropy=5.0, spread_opacity=True, scale=0.01)
    tr, _, _ = check_fused_step_vs_c_oracle(sc, 1, "weighted", f"grid_{W}x{H}")
    wm = synth.weight_map("whole", sc.gt[0]).cuda()
    tr.train_steps([0, 1, 0, 1, 1, 0], [wm] * 6)
    loss = tr.pop_loss()
    assert math.isfinite(loss) and loss > 0
    assert int(tr.tile_counts.abs().sum()) == 0 and int(tr.ticket[0]) == 0
    assert tr.overflow_events == 0


def test_knn_with_the_previous_searches_bound(env):
    """eg_knn_auto's temporal-coherence input (every point's K-th squared distance of the previous search as the
    entry bound of its list): the tables are the same with no bound, with the bound of an identical cloud, after the
    points have moved a little, and with bounds that are far too small (the re-scan path) or garbage."""
    from edgegaussians_amd import regularizers as R
    g = torch.Generator().manual_seed(9)
    n, k = 30000, 6
    pts = torch.rand(n, 3, generator=g)
    pts[: n // 4] = (pts[: n // 4] * 0.02 + 0.5)  # a dense clump inside a sparse cloud
    pts = pts.cuda()
    ref, dref = R.knn(pts, k, want_dist=True, method="grid")
    kth = torch.zeros(n, device="cuda")
    a, _ = R.knn(pts, k, method="grid", kth=kth)                      # unknown bounds (zeros) -> fills kth
    assert torch.equal(a, ref)
    assert torch.allclose(kth.sqrt(), dref[:, k - 1], rtol=1e-6)
    b, _ = R.knn(pts, k, method="grid", kth=kth)                      # tight bounds
    assert torch.equal(b, ref)
    moved = (pts + 0.002 * torch.randn(n, 3, generator=g).cuda()).contiguous()
    want, _ = R.knn(moved, k, method="grid")
    c, _ = R.knn(moved, k, method="grid", kth=kth)                    # yesterday's bounds on moved points
    assert torch.equal(c, want)
    tiny = torch.full((n,), 1e-12, device="cuda")
    d, _ = R.knn(pts, k, method="grid", kth=tiny)                     # every bound too small: re-scan without it
    assert torch.equal(d, ref)
    junk = torch.rand(n, generator=g).cuda() * 1e-3
    e, _ = R.knn(pts, k, method="grid", kth=junk)
    assert torch.equal(e, ref)


def test_operator_fast_path_equals_the_general_operator(env):
    """The drop-in's fast path for the reference's own call (one camera, colours == 1 without grad: edge_gs.py:247-268)
    -- one autograd node over the training step's kernels, cached buffers, one read-back at the end, lazy `info` --
    against the general two-node operator on the same inputs: images, every gradient, `.absgrad`, and the gsplat-layout
    `info` tensors it only computes when asked.  Non-unit colours must fall through to the general path."""
    _lib, synth, O = env
    from edgegaussians_amd import rasterizer as R
    sc = _scene(synth, n=5000, w=200, h=136, views=2)
    N, W, H = sc.means.shape[0], sc.width, sc.height
    w = synth.weight_map("weighted", sc.gt[1]).cuda()

    def run(fast, colors):
        R._FAST_ENABLED = fast
        try:
            p = [t.clone().cuda().requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
            render, alpha, info = R.rasterization(p[0], p[1], torch.exp(p[2]), torch.sigmoid(p[3]).squeeze(-1), colors,
                                                  sc.viewmats[1:2].cuda(), sc.Ks[1:2].cuda(), W, H, packed=False,
                                                  absgrad=True, rasterize_mode="antialiased")
            info["means2d"].retain_grad()  # edge_gs.py:271
            loss = (w * (torch.clamp(render[0, ..., 0], 0, 1) - sc.gt[1].cuda()).abs()).sum()
            loss.backward()
            return render.detach(), alpha.detach(), info, [t.grad for t in p], float(loss)
        finally:
            R._FAST_ENABLED = True

    ones = torch.ones(N, 3, device="cuda")
    rf, af, inf_f, gf, lf = run(True, ones)
    rg, ag, inf_g, gg, lg = run(False, ones)
    assert isinstance(inf_f, R._LazyInfo) and not isinstance(inf_g, R._LazyInfo)
    assert rf.shape == rg.shape == (1, H, W, 3) and af.shape == ag.shape == (1, H, W, 1)
    assert_close(rf, rg, rtol=1e-6, name="render") and None
    assert_close(af, ag, rtol=1e-6, name="alphas")
    assert abs(lf - lg) <= 1e-6 * abs(lg)
    for a, b, name in zip(gf, gg, ("means", "quats", "scales", "opacities")):
        assert_close(a, b, rtol=1e-5, name=f"grad {name}")
    assert_close(inf_f["means2d"].absgrad, inf_g["means2d"].absgrad, rtol=1e-5, name="absgrad")
    assert inf_f["means2d"].requires_grad and not inf_f["means2d"].is_leaf and inf_f["means2d"].shape == (1, N, 2)
    assert torch.equal(inf_f["radii"], inf_g["radii"]) and inf_f["radii"].dtype == torch.int32   # edge_gs.py:275
    assert torch.equal(inf_f["last_ids"], inf_g["last_ids"])
    for k in ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert k in inf_f and torch.equal(inf_f[k].reshape(-1), inf_g[k].reshape(-1)), k
    assert_close(inf_f["depths"], inf_g["depths"], rtol=1e-6, name="depths")
    assert_close(inf_f["conics"], inf_g["conics"], rtol=1e-6, name="conics")
    assert_close(inf_f["opacities"], inf_g["opacities"].detach(), rtol=1e-6, name="opacities")
    # colours that are not all ones: the verdict comes back with the call's one read-back and the general path takes over
    cols = torch.rand(N, 3, device="cuda")
    rc, _, inf_c, _, _ = run(True, cols)
    rc2, _, _, _, _ = run(False, cols)
    assert not isinstance(inf_c, R._LazyInfo) and torch.equal(rc, rc2)


def test_drop_in_adam_equals_torch_adam(env):
    """`edgegaussians_amd.optim.Adam` in place of `torch.optim.Adam` (train_utils.py:50-60): the same interface and state
    layout, one native launch per step.  25 steps with changing gradients and a learning-rate change through
    `param_groups` (what MultiStepLR / CustomLRScheduler do), then the reference's own state surgery (edge_gs.py:384-402:
    cull rows of the parameter and of exp_avg / exp_avg_sq, swap the parameter object) and more steps: parameters and
    moments against torch's to 2e-6 of their scale; odd sizes and a misaligned view take the scalar tail."""
    from edgegaussians_amd.optim import Adam
    g = torch.Generator().manual_seed(11)
    shapes = [(5000, 3), (5000, 4), (5000, 1), (7,), (1027, 3)]
    lrs = [2e-3, 1e-3, 1e-4, 0.03, 5e-3]
    init = [torch.randn(s, generator=g) for s in shapes]
    mk = lambda cls: [cls([torch.nn.Parameter(t.clone().cuda())], lr=lr) for t, lr in zip(init, lrs)]  # noqa: E731
    ours, theirs = mk(Adam), mk(torch.optim.Adam)
    assert all(isinstance(o, torch.optim.Optimizer) for o in ours)

    def steps(k0, k1):
        for k in range(k0, k1):
            for o, t in zip(ours, theirs):
                po, pt = o.param_groups[0]["params"][0], t.param_groups[0]["params"][0]
                gr = torch.randn(po.shape, generator=g).cuda() * (0.1 + (k % 3))
                po.grad, pt.grad = gr.clone(), gr.clone()
                if k == 12:
                    o.param_groups[0]["lr"] *= 0.1
                    t.param_groups[0]["lr"] *= 0.1
                o.step()
                t.step()
                o.zero_grad()
                t.zero_grad()

    def compare(tag):
        for i, (o, t) in enumerate(zip(ours, theirs)):
            po, pt = o.param_groups[0]["params"][0], t.param_groups[0]["params"][0]
            assert_close(po.detach(), pt.detach(), rtol=2e-6, name=f"{tag} param {i}")
            so, st = o.state[po], t.state[pt]
            assert set(so) == {"step", "exp_avg", "exp_avg_sq"} and float(so["step"]) == float(st["step"])
            assert_close(so["exp_avg"], st["exp_avg"], rtol=2e-6, name=f"{tag} exp_avg {i}")
            assert_close(so["exp_avg_sq"], st["exp_avg_sq"], rtol=2e-6, name=f"{tag} exp_avg_sq {i}")

    steps(0, 25)
    compare("25 steps")
    # the reference's remove_from_optim, applied to both
    for o in (ours[0], theirs[0]):
        param = o.param_groups[0]["params"][0]
        keep = (torch.arange(param.shape[0], device="cuda") % 3) != 0
        new = torch.nn.Parameter(param.detach()[keep].clone())
        state = o.state[param]
        del o.state[param]
        state["exp_avg"] = state["exp_avg"][keep]
        state["exp_avg_sq"] = state["exp_avg_sq"][keep]
        del o.param_groups[0]["params"][0]
        del o.param_groups[0]["params"]
        o.param_groups[0]["params"] = [new]
        o.state[new] = state
    steps(25, 31)
    compare("after the cull")
    # fused zero_grad, and a parameter whose storage is not 16-byte aligned
    base = torch.zeros(4 * 333 + 1, device="cuda")
    p = torch.nn.Parameter(base[1:])
    o = Adam([p], lr=1e-2)
    q = torch.nn.Parameter(torch.zeros(4 * 333, device="cuda"))
    t = torch.optim.Adam([q], lr=1e-2)
    for _ in range(3):
        gr = torch.randn(4 * 333, generator=g).cuda()
        p.grad, q.grad = gr.clone(), gr.clone()
        o.step(zero_grad=True)
        t.step()
        assert float(p.grad.abs().sum()) == 0.0
    assert_close(p.detach(), q.detach(), rtol=2e-6, name="misaligned")
    with pytest.raises(NotImplementedError):
        Adam([p], lr=1e-2, weight_decay=0.1)


def test_operator_protocol_with_the_drop_in_adam(env):
    """The reference's whole per-step protocol (edge_gs.py:247-279, train_gaussians.py:81-106) through the operator with
    `edgegaussians_amd.optim.Adam` against the same protocol with `torch.optim.Adam`: five steps, parameters to 1e-5."""
    _lib, synth, O = env
    from edgegaussians_amd import rasterizer as R
    from edgegaussians_amd.optim import Adam
    sc = _scene(synth, n=4000, w=200, h=136, views=3)
    N, W, H = sc.means.shape[0], sc.width, sc.height
    whole = synth.weight_map("whole", sc.gt[0]).cuda()

    def run(cls):
        P = [torch.nn.Parameter(t.clone().cuda()) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
        opts = [cls([p], lr=lr) for p, lr in zip(P, (2e-3, 1e-3, 1e-4, 0.03))]
        for s in range(5):
            v = s % 3
            render, _a, info = R.rasterization(P[0], P[1], torch.exp(P[2]), torch.sigmoid(P[3]).squeeze(-1),
                                               torch.ones(N, 3, device="cuda"), sc.viewmats[v:v + 1].cuda(),
                                               sc.Ks[v:v + 1].cuda(), W, H, packed=False, absgrad=True,
                                               rasterize_mode="antialiased")
            loss = (whole * (torch.clamp(render[0, ..., 0], 0, 1) - sc.gt[v].cuda()).abs()).sum()
            loss.backward()
            for o in opts:
                o.step()
                o.zero_grad()
        return [p.detach().clone() for p in P]

    a, b = run(Adam), run(torch.optim.Adam)
    for x, y, name in zip(a, b, ("means", "quats", "scales", "opacities")):
        assert_close(x, y, rtol=1e-5, name=name)


@pytest.mark.parametrize("size", ["tiny_grid", "small_grid", "large_grid"])
def test_item_records_follow_the_dispatch_order_contract(env, size):
    """What the wave-autonomous forward's look-back relies on (binning.hip, SegTable::slice_major / item_front / xcd_shift):
    the records the sort kernel leaves in `item_rec` that carry THIS call's tag are exactly the (tile, slice) pairs of the
    view, each once; a slice's record comes after the records of every slice in front of it in its tile (workgroups are
    dispatched in record order and only ever wait for lower records: the decoupled look-back cannot deadlock); the front
    slices come before the deeper ones; and the item numbering the hand-over storage uses (item_first / item_end /
    item_tile) stays contiguous per tile.
    Tile grids of <= 2048 tiles (round 5, XCD-aware placement): workgroup b runs on XCD b % 8, tile (tx, ty) belongs to XCD
    ((tx >> 1) + 3 (ty >> 1)) % 8, and the records of XCD x's tiles sit at indices 8 k + x, k dense from 0 -- slices [0, 4)
    of its tiles first (slices [0, 9) when the forward runs in chained mode), tile by tile, then the deeper slices; no other
    index below max_items carries the call's tag.
    Larger grids: slices [0, 9) of every tile first (the projection's scan supplies the prefix: EG_FLAG_FRONT_PREFIX), then the
    deeper slices, indices [0, n_items) without holes.  (Round 6 built the XCD-aware placement for them as well -- bands of two
    tile rows, xcd = (ty >> 1) % 8 -- and measured the forward slower with it: off, profiles/r06_xcd_large_ab.txt; the branch
    above checks it when a development build turns it on.)"""
    import numpy as np
    _lib, synth, O = env
    from edgegaussians_amd import EdgeTrainer
    # Gaussians three times the usual size: the largest tile holds several thousand (dozens of slices)
    if size == "tiny_grid":
        W, H, n_g, scale = 330, 200, 20_000, 0.012    # 21 x 13 = 273 tiles: below the XCD-aware placement's 512
    elif size == "small_grid":
        W, H, n_g, scale = 528, 400, 40_000, 0.012    # 33 x 25 = 825 tiles
    else:
        W, H, n_g, scale = 1056, 640, 130_000, 0.012  # 66 x 40 = 2640 tiles: the projection kernel scans the tiles
    sc = synth.make_scene(n_g, 2, W, H, seed=1, anisotropy=5.0, spread_opacity=True, scale=scale)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H)
    tr.ensure_capacity()
    w = synth.weight_map("whole", sc.gt[1]).cuda()
    tr.grad_step(1, w)
    torch.cuda.synchronize()
    n_items = int(tr.total.cpu()[2])
    T, tw = tr.T, (W + 15) // 16
    assert (T <= _lib.PREFIX_HERE_MAX_TILES) == (size != "large_grid")
    xcd_shift = int(_lib.load().eg_record_xcd_shift(T))
    assert xcd_shift == (1 if size == "small_grid" else 0)  # (above 2048 tiles: built, measured, off -- profiles/r06_xcd_large_ab.txt)
    table = tr.item_rec.cpu().numpy()
    valid = table[:, 2] == tr._ws_tag  # (the tag of the call just made)
    where = np.nonzero(valid)[0]
    rec = table[valid]
    first, end_ = tr.item_offsets.cpu().numpy()[:T], tr.item_end.cpu().numpy()[:T]
    item_tile = tr.item_tile.cpu().numpy()[:n_items]
    tile, sl, ns = rec[:, 0], rec[:, 1] & 0xffff, rec[:, 1] >> 16
    assert n_items > T and ns.max() >= 12, "the scene must have many-slice tiles"
    # grids of <= 2048 tiles, round 5: an EMPTY tile (other than the last one) has no record and no table entries -- its
    # sort workgroup adds the background's loss term and leaves --, but it still owns one item of the numbering
    per_tile = np.ones(T, np.int64)
    per_tile[tile] = ns
    has_rec = np.zeros(T, bool)
    has_rec[tile] = True
    if size == "large_grid":
        assert has_rec.all()
    else:
        assert has_rec[T - 1] and not has_rec.all(), "the scene must have empty tiles"
    assert len(rec) == int(per_tile[has_rec].sum()) and int(per_tile.sum()) == n_items
    # the storage numbering: contiguous runs per tile, in tile order, covering [0, n_items)
    starts = np.concatenate([[0], np.cumsum(per_tile)[:-1]])
    assert np.array_equal(first[has_rec], starts[has_rec]) and np.array_equal(end_[has_rec], (starts + per_tile)[has_rec])
    for t in np.nonzero(has_rec)[0][:: max(1, T // 64)]:
        assert (item_tile[starts[t]:starts[t] + per_tile[t]] == t).all()
    # the records: every (tile, slice) pair of the tiles that have records exactly once, consistent with the tables
    assert (sl < ns).all()
    pairs = tile.astype(np.int64) * 65536 + sl
    assert len(np.unique(pairs)) == len(rec)
    tile_end = tr.tile_end.cpu().numpy()[:T]
    assert np.array_equal(rec[:, 3], tile_end[tile])  # end of the tile's keys
    kept = tile_end[tile] - tile * tr.seg_cap
    assert np.array_equal(np.maximum(1, (kept + 127) // 128), ns)
    if size != "large_grid":
        assert (kept[tile != T - 1] > 0).all()  # (no record of an empty tile)
    # dispatch order: inside a tile by slice
    order = np.lexsort((np.arange(len(rec)), pairs))
    same_tile = tile[order][1:] == tile[order][:-1]
    assert (np.diff(where[order])[same_tile] > 0).all(), "a slice was dispatched before a slice in front of it"
    if any(k in os.environ for k in ("EG_FRONT_SLICES", "EG_FRONT_LARGE", "EG_XCD_SHIFT")):
        return
    # (the class boundary: 9 when the step's forward runs in chained mode -- a grad_step without a journal does --, 4 otherwise)
    front_small = 9 if tr._rewalk_arg(False) != -2 else 4
    if xcd_shift > 0:
        front = front_small if size == "small_grid" else 9  # kFrontChained / kFrontDefault; EG_FRONT_LARGE above 2048 tiles
        ty, tx = np.divmod(tile, tw)
        xcd = ((tx >> 1) + 3 * (ty >> 1)) % 8 if size == "small_grid" else (ty >> 1) % 8
        assert np.array_equal(where % 8, xcd), "a record sits in another XCD's list"
        assert len(np.unique(xcd)) == 8
        span = 0
        for x in range(8):
            m = xcd == x
            k = where[m] // 8
            assert np.array_equal(k, np.arange(m.sum())), "an XCD's list has a hole"
            span = max(span, 8 * int(m.sum()))
            slx, tix = sl[m], tile[m]
            n_a = int((slx < front).sum())
            # two classes (SegTable::slice_major): slices [0, front) of the list's tiles, tile by tile, then the deeper slices
            assert n_a < m.sum() and (slx[:n_a] < front).all() and (slx[n_a:] >= front).all()
            assert (np.diff(tix[:n_a]) >= 0).all() and (np.diff(tix[n_a:]) >= 0).all()
        assert span <= tr.max_items and where.max() < span
        if size == "large_grid":
            # the prefix the projection's scan left behind (ticket[1 .. T + 1]): front-class items in front of every tile
            fp = tr.ticket.cpu().numpy()
            fr = np.minimum(per_tile, front)
            assert fp[0] == 0 and np.array_equal(fp[1:T + 1], np.concatenate([[0], np.cumsum(fr)[:-1]])) and fp[T + 1] == fr.sum()
    else:
        assert np.array_equal(where, np.arange(len(rec)))  # no holes
        front = 9 if size == "large_grid" else front_small  # EG_FRONT_LARGE / the step's class boundary
        n_a = int(np.minimum(per_tile, front)[has_rec].sum())
        assert n_a < n_items and (sl[:n_a] < front).all() and (sl[n_a:] >= front).all()
        assert (np.diff(tile[:n_a]) >= 0).all() and (np.diff(tile[n_a:]) >= 0).all()  # tile by tile inside a class
        if size == "large_grid":
            # the prefix the projection's scan left behind (ticket[1 .. T + 1])
            fp = tr.ticket.cpu().numpy()
            assert fp[0] == 0 and np.array_equal(fp[1:T + 1], np.concatenate([[0], np.cumsum(np.minimum(per_tile, front))[:-1]]))
            assert fp[T + 1] == n_a
