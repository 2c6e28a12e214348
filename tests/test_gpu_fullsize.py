"""Fused step against the plain-C oracle at the sizes BASELINE.json names: configs 1, 3, 4 on synthetic look-at
poses, and the headline -- config 2 (config 5's per-GPU workload is the same shape) -- on the scan's 50 REAL poses, the
scene bench.py times, in both opacity regimes.

One view per size: gradients of `eg_train_step` (Adam off) against the C oracle's forward + backward on the
same parameters, then one whole fused step (Adam on) against `ego_train_step`.  Floats 1e-4 on EVERY element:
the Gaussians whose integer decisions are float-borderline are taken out of the scene and the borderline
pixels get zero loss weight on both sides (tests/util.py: check_fused_step_vs_c_oracle); both sets are
counted and reported (gpurun_out/parity_report.jsonl -> profiles/).
"""
import os

import pytest

from tests.util import check_fused_step_vs_c_oracle

pytestmark = pytest.mark.gpu

SIZES = {
    # name: (Gaussians, width, height, view, strategy)
    "config1": (30_000, 512, 512, 0, "whole"),
    "config3": (200_000, 1600, 1200, 1, "weighted"),
    "config4": (500_000, 1200, 680, 0, "bg_edge_ratio"),
}


@pytest.mark.parametrize("name", sorted(SIZES))
def test_fused_step_vs_c_oracle_full_size(name):
    from edgegaussians_amd import _lib, synth
    _lib.load()
    n, W, H, view, strategy = SIZES[name]
    sc = synth.make_scene(n, 2, W, H, seed=0, anisotropy=5.0, spread_opacity=True)
    check_fused_step_vs_c_oracle(sc, view, strategy, name)


REAL_POSES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cameras_00004926.npz")


@pytest.mark.parametrize("name", ["config1", "config2_real_poses", "config2_real_poses_init", "config3", "config4"])
def test_inside_the_borderline_sets_full_size(name):
    """Round 6.  The tests above remove the integer-borderline Gaussians and zero-weight the
    borderline pixels on BOTH sides; here the fused step runs on the UNCLEANED scene with the UNMASKED weight map: every
    Gaussian without a borderline pixel meets the plain 1e-4 tolerance, radius / cull decisions differ from the oracle's
    only on listed Gaussians, and a Gaussian that owns borderline pixels deviates by no more than the one-decision bound
    of those pixels (tests/util.py: check_inside_borderline_sets).  Numbers -> profiles/r06_parity_report.jsonl."""
    from edgegaussians_amd import _lib, synth
    from tests.util import check_inside_borderline_sets
    _lib.load()
    if name.startswith("config2"):
        spread = not name.endswith("init")
        sc = synth.make_scene(100_000, 50, 512, 512, seed=0, anisotropy=5.0, spread_opacity=spread, cameras_npz=REAL_POSES)
        view, strategy = (31, "bg_edge_ratio") if spread else (18, "weighted")
    else:
        n, W, H, view, strategy = SIZES[name]
        sc = synth.make_scene(n, 2, W, H, seed=0, anisotropy=5.0, spread_opacity=True)
    rec = check_inside_borderline_sets(sc, view, strategy, name)
    assert rec["gaussians_owning_a_borderline_pixel"] > 0, "the scene must exercise the threshold cases"


@pytest.mark.parametrize("n,spread,view,strategy", [(30_000, False, 0, "whole"), (30_000, True, 23, "weighted"),
                                                    (100_000, True, 12, "bg_edge_ratio")])
def test_grad_step_vs_autograd_oracle_full_size_real_poses(n, spread, view, strategy):
    """Round 5: the formulation-independent oracle -- dense PyTorch, covariance form, AUTOGRAD as the
    backward -- at FULL size: config 1 (30 k Gaussians @512x512, the scan's real poses, initial and trained-like
    opacities) and the headline config 2 (100 k).  Every full-size check so far leaned on the C oracle, whose projection
    follows the HIP kernel's factor form and whose backward is hand-derived like it: a shared derivation error in the
    projection VJP would have passed there.  Norm-wise 1e-4 on every element and the element-wise bound next to it."""
    from edgegaussians_amd import _lib, synth
    from tests.util import check_grad_step_vs_torch_oracle
    _lib.load()
    sc = synth.make_scene(n, 50, 512, 512, seed=0, anisotropy=5.0, spread_opacity=spread, cameras_npz=REAL_POSES)
    check_grad_step_vs_torch_oracle(sc, view, f"autograd_{n // 1000}k_real_poses_{'spread' if spread else 'init'}_v{view}", strategy)


@pytest.mark.parametrize("spread,view,strategy,speculate", [(True, 7, "whole", False), (True, 31, "bg_edge_ratio", False),
                                                             (False, 7, "whole", True), (False, 18, "weighted", False)])
def test_fused_step_vs_c_oracle_config2_real_poses(spread, view, strategy, speculate):
    """The headline workload itself (bench.py: 100 k Gaussians @512x512 on the real poses of scan 00004926): gradients
    and one whole Adam step against the C oracle at 1e-4 on every element -- trained-like opacities U(0.05, 0.9) (the
    chained forward resolves thousands of transmittance stops) and the initial opacity 0.08 (no pixel stops; with
    `speculate` the whole step runs the forward in its speculative mode, as the trainer does in that phase)."""
    from edgegaussians_amd import _lib, synth
    _lib.load()
    sc = synth.make_scene(100_000, 50, 512, 512, seed=0, anisotropy=5.0, spread_opacity=spread, cameras_npz=REAL_POSES)
    tr, _sc, _w = check_fused_step_vs_c_oracle(sc, view, strategy, f"config2_real_poses_{'spread' if spread else 'init'}_v{view}",
                                               speculate=speculate)
    assert tr.max_tile_seen > 1000  # the real poses put the object in the central tiles (largest tile ~2.5 k)


def test_batched_step_vs_c_oracle_config2_real_poses():
    """SURVEY 8f rank 2 pinned to the oracle directly (round 2 compared the batched step with the single-view HIP
    path only): C = 3 views of the headline scene in one launch sequence against the sum of the C oracle's per-view
    gradients and one Adam step on it."""
    from edgegaussians_amd import _lib, synth
    from tests.util import check_batched_step_vs_c_oracle
    _lib.load()
    sc = synth.make_scene(100_000, 50, 512, 512, seed=0, anisotropy=5.0, spread_opacity=True, cameras_npz=REAL_POSES)
    check_batched_step_vs_c_oracle(sc, [3, 22, 41], ["whole", "weighted", "bg_edge_ratio"], "config2_real_poses_C3")


@pytest.mark.parametrize("n,W,H,real", [(100_000, 512, 512, True), (20_000, 330, 200, False)])
def test_chained_forward_stop_records_vs_c_oracle(n, W, H, real):
    """The chained forward (exact transmittance stop inside the kernel) pinned to the oracle directly: the per-pixel
    record it leaves for the backward -- v T_final, and for every pixel whose walk stopped the id of its last
    contributor -- against the C oracle's sequential walk: the SET of stopped pixels and their last contributors
    exact outside the float-borderline pixels, T_final to 1e-5."""
    import numpy as np
    import torch
    from edgegaussians_amd import EdgeTrainer, _lib, synth
    from oracle import c_oracle as CO
    from tests.util import record, strict_inputs
    _lib.load()
    # (the small case: Gaussians three times the usual size on a 330 x 200 image with partial border tiles -- 30 % of the
    # pixels stop, the largest tile holds 3.9 k Gaussians = 31 slices)
    sc0 = synth.make_scene(n, 50 if real else 2, W, H, seed=1, anisotropy=5.0, spread_opacity=True,
                           cameras_npz=REAL_POSES if real else None, scale=0.004 if real else 0.012)
    view = 12 if real else 1
    sc, fw, border, _w, _removed = strict_inputs(sc0, view, "whole")
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H)
    tr.ensure_capacity(views=[view])
    assert tr.chained_forward
    # weights 1 everywhere and a target below every rendered value would make v = +1; simpler: a constant weight and
    # gt = -1 (render - gt > 0 on every pixel), so that the record's first word is exactly T_final
    tr.gt[view].fill_(-1.0)
    w = torch.ones(H, W, device="cuda")
    tr.grad_step(view, w)
    torch.cuda.synchronize()
    rec = tr.gtstop.cpu()
    T_hip = rec[..., 0].numpy()
    stop_id = rec[..., 1].contiguous().view(torch.int32).numpy()
    stopped_o = CO.stopped_pixels(fw)
    ok = ~border.numpy()
    # Round 5: on grids of <= 2048 tiles the fused forward leaves the pixels of EMPTY tiles (no item record) alone -- the
    # claim behind it, checked here against the oracle's per-pixel walk: nothing contributes to any pixel of such a tile
    tab = tr.item_rec.cpu().numpy()
    tiles_with_records = np.unique(tab[tab[:, 2] == tr._ws_tag][:, 0])
    tw_ = (W + 15) // 16
    has = np.zeros(tr.T, bool)
    has[tiles_with_records] = True
    pix_has = np.repeat(np.repeat(has.reshape(-1, tw_), 16, axis=0), 16, axis=1)[:H, :W]
    if tr.T <= 2560:  # (kPrefixHereMaxTiles: 2560 since round 6)
        assert not has.all(), "the scene must have empty tiles"
        assert (fw["alphas"][~pix_has] == 0).all() and not stopped_o[~pix_has].any(), "an empty tile's pixel is covered"
        ok = ok & pix_has
    T_o = 1.0 - fw["alphas"].astype(np.float64)
    covered = fw["alphas"] > 0
    # T_final (the record holds 0 where nothing contributed: T == 1)
    assert np.abs(np.where(covered, T_hip, 1.0) - T_o)[ok].max() <= 1e-5 * 1.0
    # the set of stopped pixels, and who stopped them
    assert np.array_equal((stop_id >= 0)[ok], stopped_o[ok])
    last_gauss_o = fw["flatten_ids"][np.minimum(fw["last_ids"], max(fw["M"] - 1, 0))]
    both = ok & stopped_o
    assert both.sum() > (2000 if real else 200), "the scene must make the exact stop matter"
    assert np.array_equal(stop_id[both], last_gauss_o[both])
    record("chained_forward_stop_records_vs_c_oracle", gaussians=int(sc.means.shape[0]), width=W, height=H,
           real_poses=bool(real), stopped_pixels=int(stopped_o.sum()), compared_stopped_pixels=int(both.sum()),
           borderline_pixels=int(border.sum()), stop_set_mismatches=0, last_contributor_mismatches=0,
           T_final_max_abs_err=float(np.abs(np.where(covered, T_hip, 1.0) - T_o)[ok].max()))


@pytest.mark.parametrize("name", ["config1", "config3"])
def test_operator_and_classic_layout_equal_the_fused_step_full_size(name):
    """The three ways through the library at full size, on the same inputs: the fused step (segmented binning,
    eg_train_step without Adam), the same step on the classic count / scan / emit layout, and the drop-in operator
    (`rasterization` + torch autograd, what train_gaussians.py calls): loss and every gradient tensor agree to 1e-4
    of the tensor's scale, the absgrad increment included."""
    import torch
    from edgegaussians_amd import EdgeTrainer, rasterization, synth
    from tests.util import frac_bad, record, rel_err, strict_inputs
    n, W, H, view, strategy = SIZES[name]
    sc0 = synth.make_scene(n, 2, W, H, seed=0, anisotropy=5.0, spread_opacity=True)
    sc, _fw, _border, w, _ = strict_inputs(sc0, view, strategy)
    w = w.cuda()
    N = sc.means.shape[0]
    res, losses = {}, {}
    for seg in (True, False):
        tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H,
                         segmented=seg)
        tr.ensure_capacity()
        tr.grad_step(view, w)
        res[seg] = [t.clone() for t in tr.grad_views()] + [tr.grads.view(-1)[11 * N:].clone()]
        losses[seg] = tr.pop_loss()
        del tr
    p = [t.clone().cuda().requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
    render, _alpha, info = rasterization(p[0], p[1], torch.exp(p[2]), torch.sigmoid(p[3]).squeeze(-1),
                                         torch.ones(N, 3, device="cuda"), sc.viewmats[view:view + 1].cuda(),
                                         sc.Ks[view:view + 1].cuda(), W, H, packed=False, absgrad=True,
                                         rasterize_mode="antialiased")
    loss = (w * (torch.clamp(render[0, ..., 0], 0, 1) - sc.gt[view].cuda()).abs()).sum()
    loss.backward()
    op = [p[0].grad, p[1].grad, p[2].grad, p[3].grad.view(-1), info["means2d"].absgrad[0].norm(dim=-1)]
    assert abs(losses[False] - losses[True]) <= 1e-5 * abs(losses[True])
    assert abs(float(loss.detach()) - losses[True]) <= 1e-5 * abs(losses[True])
    worst = {}
    for key, a, b, c in zip(("means", "quats", "scales", "opacities", "absgrad"), res[True], res[False], op):
        scale = float(a.abs().max())
        for other, t in (("classic", b), ("operator", c)):
            assert frac_bad(t, a, 1e-4, 1e-4 * scale) == 0.0, (name, key, other, rel_err(t, a))
            worst[f"{other}_{key}"] = rel_err(t, a)
    record("three_paths_full_size", config=name, max_norm_rel_err=max(worst.values()), by_tensor=worst)


def test_native_run_equals_single_steps_at_config3_size():
    """eg_train_steps (tail fusion across the step boundary) against single eg_train_step calls at 200 k Gaussians
    @1600x1200: above 160 k Gaussians the fused projection-backward kernel loads its operands late (the register-lean
    variant), and 7500 tiles keep the tile scan in the projection kernel -- the variants the small-scene test of
    test_gpu_parity.py does not reach."""
    import torch
    from edgegaussians_amd import EdgeTrainer, LRSchedule, synth
    from tests.util import assert_close
    n, W, H, _view, _strategy = SIZES["config3"]
    sc = synth.make_scene(n, 3, W, H, seed=0, anisotropy=5.0, spread_opacity=True)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             W, H, schedule=sched)
    ta, tb = mk(), mk()
    ta.ensure_capacity(); tb.ensure_capacity()
    views = [2, 0, 1, 2]
    wm = [synth.weight_map(("weighted", "whole")[i % 2], sc.gt[v]).cuda() for i, v in enumerate(views)]
    for v, w in zip(views, wm):
        ta.train_step(v, w)
    tb.train_steps(views, wm)
    la, lb = ta.pop_loss(), tb.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(la) and tb.overflow_events == 0 and ta.overflow_events == 0
    # bit for bit: both ways run the same arithmetic in the same order (the projection / backward / Adam code is
    # compiled without FMA contraction, so it rounds alike in the stand-alone and in the fused kernels; the forward
    # sorts by unique keys; the footprint backward reduces in a fixed order)
    for k, v in ta.state_dict().items():
        assert torch.equal(tb.state_dict()[k], v), f"run of steps: {k}"
    assert torch.equal(tb.absgrads, ta.absgrads) and torch.equal(tb.adam_m, ta.adam_m) and torch.equal(tb.adam_v, ta.adam_v)
    assert int(tb.tile_counts.abs().sum()) == 0 and int(tb.ticket[0]) == 0
