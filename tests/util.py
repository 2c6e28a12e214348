"""Comparison helpers shared by the parity tests."""
import numpy as np
import torch


def to_np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def rel_err(a, b):
    """max |a-b| / max|b| -- the norm-wise relative error the 1e-4 float tolerance is stated in."""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def frac_bad(a, b, rtol=1e-4, atol=None):
    """fraction of elements with |a-b| > atol + rtol*|b|; atol defaults to rtol * max|b|."""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    if atol is None:
        atol = rtol * max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) > atol + rtol * np.abs(b)).mean())


def assert_close(a, b, rtol=1e-4, max_bad=0.0, name="", atol_floor=0.0):
    """north_star tolerance: 1e-4 relative.  `max_bad` admits the few elements whose value hinges
    on a float-borderline branch (alpha >= 1/255, transmittance stop, radius ceil) that flips
    between two correct implementations (different exp / rounding order)."""
    # atol_floor: absolute floor for tensors that are zero up to round-off (e.g. the quaternion gradient of
    # an isotropic Gaussian), given by the caller in the units of the problem
    bn = to_np(b).astype(np.float64)
    fb = frac_bad(a, b, rtol, max(rtol * max(np.abs(bn).max() if bn.size else 0.0, 1e-30), atol_floor))
    assert fb <= max_bad, f"{name}: {fb:.2e} of elements off by > {rtol} (allowed {max_bad}); norm-rel {rel_err(a, b):.3e}"


def elementwise_report(a, b, floors=(1e-1, 1e-2, 1e-3)):
    """The element-wise statement of the float tolerance (assert_close is a max-norm check --
    |a - b| <= 1e-4 max|b| + 1e-4 |b| -- under which an element 100x below the maximum may be 1 % off).  For every floor f:
    the largest and the 99.9th-percentile RELATIVE error |a - b| / |b| over the elements with |b| >= f max|b|, and their
    number; below the smallest floor a histogram of the relative errors by decade (a sum of ~10^3 signed fp32 terms
    that cancels to 1e-3 of the largest element has lost three digits in BOTH implementations: its relative error is
    reported, not bounded)."""
    a, b = to_np(a).astype(np.float64).reshape(-1), to_np(b).astype(np.float64).reshape(-1)
    mx = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    out = {"elements": int(b.size), "max_abs_ref": float(mx)}
    for f in floors:
        sel = np.abs(b) >= f * mx
        r = rel[sel]
        out[f"ge_{f:g}"] = {"n": int(sel.sum()), "max_rel": float(r.max()) if r.size else 0.0,
                            "p999_rel": float(np.quantile(r, 0.999)) if r.size else 0.0}
    tail = rel[(np.abs(b) < min(floors) * mx) & (b != 0)]
    edges = [0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, np.inf]
    out["tail_below_smallest_floor"] = {"n": int(tail.size), "hist_rel_err_decades_from_1e-6": np.histogram(tail, bins=edges)[0].tolist()}
    nz = int(((b == 0) & (a != 0)).sum())
    out["nonzero_where_ref_is_zero"] = nz
    return out


# The element-wise statement of the tolerance, asserted next to the norm-wise one on every gradient comparison: the
# RELATIVE error of an element is bounded by a tier that follows its size against the largest element of its tensor --
#     |ref| >= 0.1   max|ref| : 1e-4   (north_star's figure holds element by element here; measured <= 8.2e-5)
#     |ref| >= 0.01  max|ref| : 6e-4   (measured <= 4.5e-4; round 5 allowed 1e-3)
#     |ref| >= 0.001 max|ref| : 5e-3   (measured <= 3.8e-3; round 5 allowed 1e-2)
# i.e. an ABSOLUTE error of ~1e-5 max|ref| (measured <= 8e-6), ten times tighter than assert_close's 1e-4 max|ref|: a
# gradient element is a sum of ~10^3 signed fp32 terms of the size of the large elements, and one that cancels to 1e-3
# of them has lost three digits in both implementations.  Round 6 tightened the two lower tiers to what three rounds of
# measurements support (worst cases: the quaternion gradient at 200 k Gaussians @1600x1200, profiles/r06_parity_report.jsonl);
# 1e-4 at 0.01 max|ref| is NOT supported by them (4.5e-4 measured), which is the cancellation above, not a defect: the same
# elements agree to 8e-6 of the maximum.  Below 1e-3 max|ref| the relative errors are reported as a histogram, not bounded.
# EG_ELEMENTWISE_SCALE loosens all tiers (recording runs).
ELEMENTWISE_TIERS = ((1e-1, 1e-4), (1e-2, 6e-4), (1e-3, 5e-3))
_EW_SCALE = float(__import__("os").environ.get("EG_ELEMENTWISE_SCALE", "1"))


def assert_elementwise(got, ref, keys, label):
    rep = {k: elementwise_report(got[k], ref[k], floors=tuple(f for f, _ in ELEMENTWISE_TIERS)) for k in keys}
    for k in keys:
        for floor, rtol in ELEMENTWISE_TIERS:
            worst = rep[k][f"ge_{floor:g}"]["max_rel"]
            assert worst <= rtol * _EW_SCALE, (f"{label} grad {k}: element-wise relative error {worst:.3e} > {rtol} on an "
                                               f"element with |ref| >= {floor} max|ref|")
    return rep


def check_grad_step_vs_torch_oracle(sc, view, label, strategy="whole", clean=True):
    """fused eg_train_step (no Adam) against the dense PyTorch oracle's AUTOGRAD -- a backward derived independently of
    every hand-written one (the C oracle's backward is hand-derived like the HIP kernels') -- on the same inputs; 1e-4
    on every element in the norm-wise sense, the element-wise bound of assert_elementwise next to it.  Any size the
    tile-chunked oracle holds in memory (30 k Gaussians @512x512: ~3 s and 2.4 GB on CPU).  Returns the stopped share."""
    from edgegaussians_amd import EdgeTrainer
    from oracle import ref_torch as O
    sc, fw, border, w, removed = strict_inputs(sc, view, strategy)
    N = sc.means.shape[0]
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,
                     sc.width, sc.height)
    tr.ensure_capacity(views=[view])
    tr.grad_step(view, w.cuda())
    names = ("means", "quats", "scales", "opac", "absgrad")
    got = dict(zip(names, [t.clone().cpu() for t in tr.grad_views()] + [tr.grads.view(-1)[11 * N:].clone().cpu()]))
    loss_g = tr.pop_loss()
    assert not tr.overflowed()
    p = [t.clone().requires_grad_(True) for t in (sc.means, sc.quats, sc.log_scales, sc.logit_opacities)]
    render, alpha, info = O.rasterization(
        means=p[0], quats=p[1], scales=torch.exp(p[2]), opacities=torch.sigmoid(p[3]).squeeze(-1),
        colors=torch.ones(N, 3), viewmats=sc.viewmats[view:view + 1], Ks=sc.Ks[view:view + 1], width=sc.width,
        height=sc.height, packed=False, absgrad=True, rasterize_mode="antialiased")
    info["means2d"].retain_grad()
    loss = O.edge_step_loss(render[0, ..., 0], sc.gt[view], w)
    loss.backward()
    assert abs(loss_g - float(loss)) <= 1e-4 * abs(float(loss)), (loss_g, float(loss))
    want = dict(zip(names, [p[0].grad, p[1].grad, p[2].grad, p[3].grad.view(-1), info["means2d"].absgrad[0].norm(dim=-1)]))
    stopped = float((alpha.detach() > 1 - 1.1e-4).float().mean())
    errs = {k: rel_err(got[k], want[k]) for k in names}
    for k in names:
        assert_close(got[k], want[k], rtol=1e-4, name=f"{label} {k}")
    rep = assert_elementwise(got, want, names, label)
    record("fused_grad_step_vs_torch_oracle", scene=label, gaussians=N, removed_borderline_gaussians=removed,
           borderline_pixels=int(border.sum()), pixels=int(border.numel()), stopped_pixel_frac=stopped, max_rel_err=errs,
           elementwise=rep)
    return stopped


def filter_fixture(golden_dir):
    """Inputs of the reference's filter_by_projection run (tests/golden/make_golden.py:filter_projection):
    means, float edge images (DexiNed / 255) and the camera dicts of filtering.py:42-56."""
    import os
    import numpy as np
    d = np.load(os.path.join(golden_dir, "filter_projection.npz"))
    cams = np.load(os.path.join(golden_dir, "cameras_00004926.npz"))
    edges = np.load(os.path.join(golden_dir, "edges_00004926.npz"))
    H, W = int(cams["height"]), int(cams["width"])
    images, cameras = [], []
    for k in d["views"]:
        im = np.zeros(H * W, np.float32)
        im[edges[f"idx_{k}"]] = edges[f"val_{k}"].astype(np.float32) / np.float32(255.0)
        images.append(im.reshape(H, W))
        vm = cams["viewmats"][k]
        cameras.append({"K": cams["Ks"][k], "R": vm[:3, :3], "t": vm[:3, 3:], "h": H, "w": W})
    return d, images, cameras


# ---------------------------------------------------------------------------------------------------
# Quantified float-borderline sets (oracle/eg_oracle.c: ego_project_borderline, ego_borderline_pixels).
# The path branches on float comparisons; two correct fp32 implementations may branch differently where
# the compared value is within rounding distance of its threshold.  Instead of tolerating an unexplained
# fraction of outliers, the parity tests (a) take the Gaussians whose INTEGER decisions (radius ceil,
# culls, tile box) are borderline out of the scene, (b) give the pixels whose walk passes within a stated
# margin of a threshold (alpha 1/255, alpha 0.999, T 1e-4, sign of render - gt) ZERO loss weight -- in both
# implementations -- and then assert the 1e-4 tolerance on EVERY element.  Both sets are computed in double
# precision from the oracle's own fp32 values and their sizes are asserted small and reported.
REL_GAUSS = 2e-5   # relative margin on radius ceil / culls / tile box (fp32 chain error ~1e-6)
REL_ALPHA = 1e-4   # relative margin on alpha thresholds (projection differences move alpha by ~1e-5)
REL_T = 2e-4       # relative margin on the transmittance stop (accumulated product error ~1e-5)


def clean_scene(sc, views, rel=REL_GAUSS):
    """Returns (scene without the Gaussians that are integer-borderline in any of `views`, how many went)."""
    import dataclasses
    from oracle import c_oracle as CO
    scales = torch.exp(sc.log_scales).numpy()
    bad = np.zeros(sc.means.shape[0], bool)
    for v in views:
        bad |= CO.project_borderline(sc.means.numpy(), sc.quats.numpy(), scales, sc.viewmats[v].numpy(),
                                     sc.Ks[v].numpy(), sc.width, sc.height, rel=rel) > 0
    keep = torch.from_numpy(~bad)
    out = dataclasses.replace(sc, means=sc.means[keep].contiguous(), log_scales=sc.log_scales[keep].contiguous(),
                              quats=sc.quats[keep].contiguous(), logit_opacities=sc.logit_opacities[keep].contiguous())
    return out, int(bad.sum())


def oracle_forward(sc, view, params=None):
    """C-oracle forward of one view (numpy dict, see oracle/c_oracle.py:rasterize) at the scene's -- or
    the given raw -- parameters, unit colours, antialiased."""
    from oracle import c_oracle as CO
    means, ls, q, lo = params if params is not None else (sc.means, sc.log_scales, sc.quats, sc.logit_opacities)
    n = means.shape[0]
    return CO.rasterize(to_np(means), to_np(q), np.exp(to_np(ls).astype(np.float32)),
                        (1.0 / (1.0 + np.exp(-to_np(lo).astype(np.float32).reshape(-1)))).astype(np.float32),
                        np.ones((n, 1), np.float32), to_np(sc.viewmats[view]), to_np(sc.Ks[view]), sc.width, sc.height)


def borderline_pixel_mask(fw, gt=None, rel_alpha=REL_ALPHA, rel_T=REL_T):
    """bool [H,W] from a C-oracle forward: pixels within the margins of a float threshold; with `gt` also the
    pixels where the sign of clamp(render) - gt (the L1 gradient) hinges on rounding."""
    from oracle import c_oracle as CO
    m = CO.borderline_pixels(fw, rel_alpha, rel_T) > 0
    if gt is not None:
        d = np.clip(fw["render"][..., 0], 0.0, 1.0) - to_np(gt)
        m |= (np.abs(d) < 1e-6) & (d != 0)
    return torch.from_numpy(m)


def masked_weights(w, mask):
    w = w.clone()
    w[mask] = 0.0
    return w


def record(name, **metrics):
    """Appends one line to gpurun_out/parity_report.jsonl: the MEASURED errors behind the assertions."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v)
                                              for k, v in metrics.items()}}) + "\n")


def record_cpu(name, **metrics):
    """The same for the CPU-only tests (oracle against oracle): profiles-bound lines in gpurun_out/parity_report_cpu.jsonl."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report_cpu.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v)
                                              for k, v in metrics.items()}}) + "\n")


def strict_inputs(sc, view, strategy, seed=3, ratio=1.0):
    """(clean scene, C-oracle forward of `view`, borderline pixel mask, masked weight map, #removed Gaussians)."""
    from edgegaussians_amd import synth
    sc2, removed = clean_scene(sc, [view])
    fw = oracle_forward(sc2, view)
    border = borderline_pixel_mask(fw, sc2.gt[view])
    w = masked_weights(synth.weight_map(strategy, sc2.gt[view], ratio, torch.Generator().manual_seed(seed)), border)
    return sc2, fw, border, w, removed


def oracle_raw_grads(sc, fw, w, view):
    """Loss and gradients w.r.t. the RAW parameters (means, quats, log-scales, logit-opacities) + the absgrad
    increment, from the C oracle's forward `fw` and its sequential backward; chain rule in float64."""
    from oracle import c_oracle as CO
    gt = sc.gt[view]
    d = torch.clamp(torch.from_numpy(fw["render"][..., 0]), 0, 1) - gt
    loss = float((w.double() * d.abs().double()).sum())
    want = CO.backward(fw, (w * torch.sign(d)).numpy()[..., None].astype(np.float32))
    scales64 = np.exp(sc.log_scales.numpy().astype(np.float64))
    op64 = 1.0 / (1.0 + np.exp(-sc.logit_opacities.numpy().astype(np.float64).reshape(-1)))
    return loss, {"means": want["means"], "quats": want["quats"], "scales": want["scales"] * scales64,
                  "opac": want["opacities"] * op64 * (1.0 - op64),
                  "absgrad": np.linalg.norm(want["absgrad"].astype(np.float64), axis=1),
                  "means2d": want["means2d"], "absgrad2": want["absgrad"]}


def check_fused_step_vs_c_oracle(sc, view, strategy, label, trainer_kwargs=None, whole_step=True, speculate=False):
    """eg_train_step (Adam off, then Adam on) against the C oracle on one view: every float at 1e-4, integer
    bookkeeping exact; borderline Gaussians removed, borderline pixels zero-weighted, both counted + reported.
    speculate: the whole step runs the forward in its SPECULATIVE mode (no exact-stop handling: what the trainer
    does while no pixel reaches the transmittance stop) and must not have needed a replay.
    Returns the trainer for further checks."""
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    from oracle import c_oracle as CO
    n0 = sc.means.shape[0]
    sc, fw, border, w, removed = strict_inputs(sc, view, strategy)
    N, W, H = sc.means.shape[0], sc.width, sc.height
    # (caps at 2x the largest sets measured over the sizes and seeds of profiles/r04_parity_report.jsonl: 0.8 % of the
    # Gaussians, 0.3 % of the pixels)
    assert removed <= max(3, 0.016 * n0) and float(border.float().mean()) < 0.006, (removed, n0, float(border.float().mean()))
    loss_o, ref = oracle_raw_grads(sc, fw, w, view)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H,
                     schedule=sched, **(trainer_kwargs or {}))
    tr.ensure_capacity(views=[view])
    tr.grad_step(view, w.cuda().contiguous())
    gm, gq, gs, go = [t.clone().cpu() for t in tr.grad_views()]
    got = {"means": gm, "quats": gq, "scales": gs, "opac": go, "absgrad": tr.grads.view(-1)[11 * N:].clone().cpu()}
    loss_g = tr.pop_loss()
    m_tight = tr.last_m()
    assert not tr.overflowed()
    assert (0 < m_tight <= fw["M"]) or fw["M"] == 0  # tight tile boxes only ever drop (Gaussian, tile) pairs
    assert abs(loss_g - loss_o) <= 1e-4 * abs(loss_o), (loss_g, loss_o)
    keys = ("means", "quats", "scales", "opac", "absgrad")
    errs = {k: rel_err(got[k], ref[k]) for k in keys}
    record("fused_grad_step_vs_c_oracle", size=label, gaussians=N, removed_borderline_gaussians=removed,
           borderline_pixels=int(border.sum()), pixels=int(border.numel()), M_gsplat_boxes=int(fw["M"]), M_tight=m_tight,
           loss_rel_err=abs(loss_g - loss_o) / abs(loss_o), grad_max_rel_err=errs,
           stopped_pixel_frac=float((torch.from_numpy(fw["alphas"]) > 1 - 1.1e-4).float().mean()))
    for k in keys:
        assert_close(got[k], ref[k], rtol=1e-4, name=f"{label} grad {k}")
    # ... and element-wise (round 5): every gradient element with |ref| >= 1e-3 max|ref|, the tail below it as a histogram
    record("fused_grad_step_vs_c_oracle_elementwise", size=label, gaussians=N, tiers=ELEMENTWISE_TIERS,
           elementwise=assert_elementwise(got, ref, keys, label))
    if not whole_step:
        return tr, sc, w
    # ---- one whole fused step (forward + loss + backward + absgrad + Adam) against ego_train_step
    ct = CO.CpuTrainer(sc.means.numpy(), sc.log_scales.numpy(), sc.quats.numpy(), sc.logit_opacities.numpy(), sched.at(0))
    lc, M = ct.train_step(sc.viewmats[view].numpy(), sc.Ks[view].numpy(), W, H, sc.gt[view].numpy(), w.numpy())
    if speculate:
        tr.rewalk_hint = 0  # "no pixel stopped lately": the journalled step speculates on it
        tr._args_cache = {}
        from edgegaussians_amd import _lib
        assert tr._rewalk_arg(True) == _lib.REWALK_SPECULATE
    tr.train_step(view, w.cuda().contiguous())
    lg = tr.pop_loss()
    assert not speculate or tr.rewalk_misses == 0, "the speculative forward saw a stop: pick a scene without stops"
    assert abs(lg - lc) <= 1e-4 * abs(lc) and M == fw["M"] and not tr.overflowed()
    assert_close(tr.absgrads, ct.absgrads, rtol=1e-4, name=f"{label} absgrads")
    # first Adam step: delta = -lr g / (|g| + eps).  Adam's epsilon makes the step ill-conditioned where
    # |g| ~ eps: a gradient error dg moves delta by lr eps dg / (|g| + eps)^2.  With dg <= 1e-4 max|g| (the
    # float tolerance asserted on the gradients above; the fused kernel's own gradient differs from grad_step's
    # by rounding of that order -- another template instantiation, another FMA contraction) the PROPAGATED
    # tolerance of element i is   lr (1e-4 + eps 1e-4 max|g| / (|g_i| + eps)^2) + one ulp of the parameter,
    # asserted on EVERY element, (a) against the C oracle's step and (b) against Adam's first step evaluated in
    # float64 on the gradient the HIP path produced.
    lrs = sched.at(0)
    derr, aerr, worst_bound = {}, {}, {}
    for key, mine, theirs, init, lr in (("means", tr.means, ct.means, sc.means, lrs["means"]),
                                        ("scales", tr.log_scales, ct.log_scales, sc.log_scales, lrs["scales"]),
                                        ("quats", tr.quats, ct.quats, sc.quats, lrs["quats"]),
                                        ("opac", tr.logit_opacities, ct.logit, sc.logit_opacities.view(-1), lrs["opacities"])):
        g = torch.from_numpy(np.abs(np.asarray(ref[key], dtype=np.float64)).reshape(-1))
        ulp = float(init.abs().max()) * 6e-8  # the parameter itself is rounded to fp32 after the update
        bound = lr * (1e-4 + 1e-8 * (1e-4 * float(g.max())) / (g + 1e-8) ** 2) + ulp
        have = mine.cpu().reshape(-1).double() - init.reshape(-1).double()
        dc = torch.from_numpy(theirs).reshape(-1).double() - init.reshape(-1).double()
        gg = got[key].double().reshape(-1)
        want_delta = -(lr / (1.0 - 0.9)) * (0.1 * gg) / (torch.sqrt(0.001 * gg * gg) / np.sqrt(1.0 - 0.999) + 1e-8)
        derr[key] = float(((have - dc).abs() / (2 * bound)).max())      # both sides carry gradient rounding
        aerr[key] = float(((have - want_delta).abs() / bound).max())
        worst_bound[key] = float(bound.max() / lr)
        assert derr[key] <= 1.0, f"{label} delta {key} vs C oracle: {derr[key]} x the propagated tolerance"
        assert aerr[key] <= 1.0, f"{label} delta {key} vs float64 Adam: {aerr[key]} x the propagated tolerance"
    record("fused_train_step_vs_c_oracle", size=label, loss_rel_err=abs(lg - lc) / abs(lc),
           absgrads_max_rel_err=rel_err(tr.absgrads, ct.absgrads), adam_delta_err_over_tolerance_vs_c_oracle=derr,
           adam_delta_err_over_tolerance_vs_float64_adam=aerr, largest_propagated_tolerance_in_lr=worst_bound)
    return tr, sc, w


def check_batched_step_vs_c_oracle(sc, views, strategies, label):
    """eg_train_step_batched (C views per launch sequence, ONE optimizer step) against the C oracle: the batched
    gradient buffer = sum over the views of the oracle's per-view gradients (1e-4 on every element), the loss = sum of
    the oracle's losses, and the whole batched step = Adam's first step (float64) on that sum with the propagated
    tolerance of check_fused_step_vs_c_oracle.  Borderline Gaussians of ANY of the views leave the scene, each view's
    borderline pixels get zero weight in that view."""
    from edgegaussians_amd import EdgeTrainer, LRSchedule, synth
    n0 = sc.means.shape[0]
    sc, removed = clean_scene(sc, list(views))
    N, W, H = sc.means.shape[0], sc.width, sc.height
    assert removed <= max(3, 0.03 * n0)
    wm, loss_o, ref, n_border = [], 0.0, None, 0
    for k, (v, strat) in enumerate(zip(views, strategies)):
        fw = oracle_forward(sc, v)
        border = borderline_pixel_mask(fw, sc.gt[v])
        n_border += int(border.sum())
        w = masked_weights(synth.weight_map(strat, sc.gt[v], 1.0, torch.Generator().manual_seed(3 + k)), border)
        lo, r = oracle_raw_grads(sc, fw, w, v)
        loss_o += lo
        ref = r if ref is None else {key: ref[key] + r[key] for key in ref}
        wm.append(w.cuda().contiguous())
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H, schedule=sched)
    tr.ensure_capacity(views=list(views))
    tr.grad_step_batched(list(views), wm)
    gm, gq, gs, go = [t.clone().cpu() for t in tr.grad_views()]
    got = {"means": gm, "quats": gq, "scales": gs, "opac": go, "absgrad": tr.grads.view(-1)[11 * N:].clone().cpu()}
    loss_g = tr.pop_loss()
    assert not tr.overflowed() and abs(loss_g - loss_o) <= 1e-4 * abs(loss_o), (loss_g, loss_o)
    keys = ("means", "quats", "scales", "opac", "absgrad")
    errs = {k: rel_err(got[k], ref[k]) for k in keys}
    for k in keys:
        assert_close(got[k], ref[k], rtol=1e-4, name=f"{label} batched grad {k}")
    # the whole batched step: one Adam step on the summed gradient
    tr.train_step_batched(list(views), wm)
    lg = tr.pop_loss()
    assert abs(lg - loss_o) <= 1e-4 * abs(loss_o) and not tr.overflowed() and tr.overflow_events == 0
    assert_close(tr.absgrads, ref["absgrad"], rtol=1e-4, name=f"{label} batched absgrads")
    lrs = sched.at(0)
    aerr = {}
    for key, mine, init, lr in (("means", tr.means, sc.means, lrs["means"]), ("scales", tr.log_scales, sc.log_scales, lrs["scales"]),
                                ("quats", tr.quats, sc.quats, lrs["quats"]),
                                ("opac", tr.logit_opacities, sc.logit_opacities.view(-1), lrs["opacities"])):
        g = torch.from_numpy(np.abs(np.asarray(ref[key], dtype=np.float64)).reshape(-1))
        ulp = float(init.abs().max()) * 6e-8
        bound = lr * (1e-4 + 1e-8 * (1e-4 * float(g.max())) / (g + 1e-8) ** 2) + ulp
        have = mine.cpu().reshape(-1).double() - init.reshape(-1).double()
        gg = torch.from_numpy(np.asarray(ref[key], dtype=np.float64).reshape(-1))
        want_delta = -(lr / (1.0 - 0.9)) * (0.1 * gg) / (torch.sqrt(0.001 * gg * gg) / np.sqrt(1.0 - 0.999) + 1e-8)
        aerr[key] = float(((have - want_delta).abs() / (2 * bound)).max())
        assert aerr[key] <= 1.0, f"{label} batched delta {key} vs float64 Adam on the oracle's sum: {aerr[key]} x tolerance"
    record("batched_step_vs_c_oracle", size=label, views=len(views), gaussians=N, removed_borderline_gaussians=removed,
           borderline_pixels=n_border, loss_rel_err=abs(loss_g - loss_o) / abs(loss_o), grad_max_rel_err=errs,
           adam_delta_err_over_tolerance=aerr)
    return tr


# ---------------------------------------------------------------------------------------------------
# Round 6: what the HIP path does INSIDE the borderline sets.  check_fused_step_vs_c_oracle takes
# the integer-borderline Gaussians out of the scene and gives the borderline pixels zero weight ON BOTH SIDES: a kernel
# that mishandled exactly the threshold cases would pass it.  This check runs the SAME step on the UNCLEANED scene with
# the UNMASKED weight map and asserts, element by element,
#   * outside the attributable set: the plain 1e-4 tolerance (so the masking only ever protected attributable rows);
#   * integer decisions (radius, culls) differ from the oracle's ONLY on Gaussians the oracle lists as borderline;
#   * inside: |got - ref| <= the 1e-4 tolerance + the ONE-DECISION bound of the Gaussian's borderline pixels.
# One-decision bound.  With unit colours the 2-D gradient of Gaussian g is a sum over the pixels p it contributes to of
# v_p T_final(p) / (1 - alpha_g) * d alpha_g / d theta  (DESIGN.md section 5), and T_final / (1 - alpha_g) <= T_before(g):
# whatever two implementations decide at a pixel whose walk passes within the margins of a threshold (alpha 1/255, alpha
# 0.999, T 1e-4) -- include the Gaussian or not, stop there or one later -- each one's term of that pixel is bounded by
# c_{g,p} = |v_p| T_before(g, p) |d alpha_g / d theta| evaluated as if included, both terms have the sign of v_p, so they
# differ by at most c_{g,p}; where the SIGN of v_p itself hinges on rounding (|render - gt| < 1e-6) by 2 c_{g,p}.
# T_before comes from the oracle's walk (+1 %: a flip in front of g moves it by <= 1/255 + margin).  The bound of a
# Gaussian is the sum over ITS borderline pixels -- typically 1-3 of the hundreds it covers -- and zero for everybody else.
def _borderline_pixel_bounds(fw, w, gt, border, alpha_floor=(1.0 / 255.0) * (1.0 - 4 * REL_ALPHA)):
    """[N, 8] one-decision bounds (columns of g2d: v_x v_y |v_x| |v_y| v_a v_b v_c v_o) and the bool [N] set of Gaussians
    that own a weighted borderline pixel, from a C-oracle forward `fw` (float64 arithmetic on its fp32 values)."""
    width, height = fw["_size"]
    tw = (width + 15) // 16
    N = fw["means2d"].shape[0]
    bound = np.zeros((N, 8), np.float64)
    owns = np.zeros(N, bool)
    offs = fw["isect_offsets"].reshape(-1).astype(np.int64)
    ends = np.concatenate([offs[1:], [fw["M"]]])
    flat = fw["flatten_ids"]
    m2d, con, op = fw["means2d"].astype(np.float64), fw["conics"].astype(np.float64), fw["opacities"].astype(np.float64)
    d_img = np.clip(fw["render"][..., 0], 0.0, 1.0).astype(np.float64) - to_np(gt).astype(np.float64)
    wn = to_np(w).astype(np.float64)
    ys, xs = np.nonzero(to_np(border) & (wn != 0))
    for py, px in zip(ys, xs):
        t = (py // 16) * tw + px // 16
        ids = flat[offs[t]:ends[t]]
        if ids.size == 0:
            continue
        dx, dy = m2d[ids, 0] - (px + 0.5), m2d[ids, 1] - (py + 0.5)
        a, b, c = con[ids, 0], con[ids, 1], con[ids, 2]
        sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
        vis = np.exp(-np.maximum(sigma, 0.0))
        araw = op[ids] * vis
        cand = (sigma >= -1e-6) & (araw >= alpha_floor)            # may be included by SOME correct implementation
        alpha = np.where(cand, np.minimum(0.999, araw), 0.0)
        t_before = np.concatenate([[1.0], np.cumprod(1.0 - alpha)[:-1]]) * 1.01
        kappa = 2.0 if (abs(d_img[py, px]) < 1e-6 and d_img[py, px] != 0) else 1.0
        v = kappa * abs(wn[py, px]) * t_before * cand
        wm = v * araw                                                # |dL/dsigma| as if included
        gx, gy = np.abs(a * dx + b * dy), np.abs(b * dx + c * dy)
        rows = np.stack([wm * gx, wm * gy, wm * gx, wm * gy, 0.5 * wm * dx * dx, wm * np.abs(dx * dy), 0.5 * wm * dy * dy,
                         v * vis], axis=1)
        np.add.at(bound, ids, rows)
        owns[ids[cand]] = True
    return bound, owns


def check_inside_borderline_sets(sc, view, strategy, label, trainer_kwargs=None, seed=3, ratio=1.0):
    """See the comment above.  Returns the record written to parity_report.jsonl."""
    from edgegaussians_amd import EdgeTrainer, synth
    from oracle import c_oracle as CO
    N, W, H = sc.means.shape[0], sc.width, sc.height
    fw = oracle_forward(sc, view)
    border = borderline_pixel_mask(fw, sc.gt[view])
    w = synth.weight_map(strategy, sc.gt[view], ratio, torch.Generator().manual_seed(seed))   # UNMASKED
    listed = CO.project_borderline(sc.means.numpy(), sc.quats.numpy(), torch.exp(sc.log_scales).numpy(), sc.viewmats[view].numpy(),
                                   sc.Ks[view].numpy(), W, H, rel=REL_GAUSS) > 0
    loss_o, ref = oracle_raw_grads(sc, fw, w, view)
    want = CO.backward(fw, (w * torch.sign(torch.clamp(torch.from_numpy(fw["render"][..., 0]), 0, 1) - sc.gt[view])).numpy()[..., None].astype(np.float32))
    ref2d = np.concatenate([want["means2d"], want["absgrad"], want["conics"], want["opacities_eff"][:, None]], axis=1).astype(np.float64)
    tr = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt, W, H, **(trainer_kwargs or {}))
    tr.ensure_capacity(views=[view])
    tr.grad_step(view, w.cuda().contiguous())
    gm, gq, gs, go = [t.clone().cpu() for t in tr.grad_views()]
    got = {"means": gm, "quats": gq, "scales": gs, "opac": go, "absgrad": tr.grads.view(-1)[11 * N:].clone().cpu()}
    assert tr.ref_index is None, "rows in the scene's order (spatial_order off)"
    g2d = tr.g2d.clone().cpu()
    radii_g = tr.splat.view(-1, 8)[:, 7].clone().view(torch.int32).cpu()
    loss_g = tr.pop_loss()
    assert not tr.overflowed()
    # ---- integer decisions: a radius (0 = culled) that differs from the oracle's only on a LISTED borderline Gaussian
    rad_o = fw["radii"]
    flipped = radii_g.numpy() != rad_o
    assert not (flipped & ~listed).any(), (f"{label}: {int((flipped & ~listed).sum())} Gaussians take another radius / cull decision than the "
                                           "oracle WITHOUT being float-borderline")
    # everybody whose 3-sigma box overlaps a flipped Gaussian's (either radius) shares pixels with it: attributable to it
    near_flip = np.zeros(N, bool)
    r_any = np.maximum(radii_g.numpy(), rad_o).astype(np.float64)
    mx, my = fw["means2d"][:, 0].astype(np.float64), fw["means2d"][:, 1].astype(np.float64)
    for f in np.nonzero(flipped)[0]:
        near_flip |= (np.abs(mx - mx[f]) <= r_any + r_any[f]) & (np.abs(my - my[f]) <= r_any + r_any[f])
    near_flip |= flipped
    # ---- the 2-D gradients, every Gaussian, every component
    bound, owns = _borderline_pixel_bounds(fw, w, sc.gt[view], border)
    vis = rad_o > 0
    dev = np.abs(g2d.numpy().astype(np.float64) - ref2d)
    cols = ((0, 2), (2, 4), (4, 7), (7, 8))
    tol = np.zeros_like(dev)
    for c0, c1 in cols:   # (assert_close's criterion, per block of like components)
        tol[:, c0:c1] = 1e-4 * np.abs(ref2d[vis][:, c0:c1]).max() + 1e-4 * np.abs(ref2d[:, c0:c1])
    rows = vis & ~near_flip
    over = dev[rows] - tol[rows] - bound[rows]
    worst = float((dev[rows] / (tol[rows] + bound[rows])).max())
    assert (over <= 0).all(), (f"{label}: 2-D gradient off by {worst:.3f} x (1e-4 tolerance + one-decision bound of the Gaussian's borderline "
                               f"pixels) on {int((over > 0).any(axis=1).sum())} Gaussians")
    plain = rows & ~owns
    assert (dev[plain] <= tol[plain]).all(), f"{label}: a Gaussian WITHOUT a borderline pixel misses the plain 1e-4 tolerance"
    needed = (dev[rows & owns] > tol[rows & owns]).any(axis=1)     # rows where the bound was actually used
    use = dev[rows & owns][needed] / np.maximum(bound[rows & owns][needed], 1e-300)
    # ---- raw-parameter gradients of everybody not attributable: the plain tolerance, unmasked weights, uncleaned scene
    clear = ~near_flip & ~owns
    keys = ("means", "quats", "scales", "opac", "absgrad")
    raw_err = {}
    for k in keys:
        a_, b_ = to_np(got[k]).astype(np.float64).reshape(N, -1), np.asarray(ref[k], np.float64).reshape(N, -1)
        t_ = 1e-4 * np.abs(b_).max() + 1e-4 * np.abs(b_)
        raw_err[k] = float((np.abs(a_ - b_)[clear] / t_[clear]).max()) if clear.any() else 0.0
        assert raw_err[k] <= 1.0, f"{label} raw grad {k}: {raw_err[k]:.2f} x the 1e-4 tolerance on a Gaussian with no borderline pixel"
    # ---- the loss: a borderline pixel moves |render - gt| by at most one threshold contribution
    loss_slack = float((to_np(w).astype(np.float64) * to_np(border)).sum()) * (1.0 / 255.0 + 2e-4)
    assert abs(loss_g - loss_o) <= 1e-4 * abs(loss_o) + loss_slack, (loss_g, loss_o, loss_slack)
    rec = dict(size=label, gaussians=N, listed_borderline_gaussians=int(listed.sum()), radius_or_cull_flips=int(flipped.sum()),
               rows_near_a_flip=int(near_flip.sum()), borderline_pixels=int(border.sum()), pixels=int(border.numel()),
               gaussians_owning_a_borderline_pixel=int((owns & vis).sum()), rows_that_needed_the_bound=int(needed.sum()),
               worst_dev_over_tolerance_plus_bound=worst, worst_dev_over_bound_where_needed=float(use.max()) if use.size else 0.0,
               raw_grad_err_over_tolerance_outside=raw_err, loss_rel_err=abs(loss_g - loss_o) / abs(loss_o))
    record("inside_borderline_sets", **rec)
    return rec
