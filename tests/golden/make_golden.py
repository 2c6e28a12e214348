#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the importable parts of the reference.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from ``/root/reference`` (read-only, nothing is copied): ``edgegaussians.models.
edge_gs``, ``.models.losses``, ``.utils.misc_utils``, ``.utils.train_utils``, ``.cameras.cameras``,
``.data.dataparsers``.  Modules the image lacks (ipdb, open3d, plyfile, dacite, tensorboard,
gsplat) are stubbed in ``sys.modules``; ``gsplat.rasterization`` is a RECORDING stub, so the
boundary trace is exactly what the reference's model class passes / reads back
(edge_gs.py:250-275).  Outputs are data only (inputs + expected outputs).
"""
import dataclasses
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SCAN = os.path.join(REF, "data/ABC-NEF_Edge/data/00004926")
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _from_dict(data_class, data):
    names = {f.name for f in dataclasses.fields(data_class)}
    return data_class(**{k: v for k, v in data.items() if k in names})  # dacite non-strict


TRACE = {}


def _recording_rasterization(**kw):
    TRACE["kwargs"] = {
        k: ({"shape": list(v.shape), "dtype": str(v.dtype), "requires_grad": bool(v.requires_grad)}
            if isinstance(v, torch.Tensor) else v)
        for k, v in kw.items()
    }
    N = kw["means"].shape[0]
    C = kw["viewmats"].shape[0]
    H, W = kw["height"], kw["width"]
    means2d = (kw["means"][None, :, :2] * 1.0).expand(C, N, 2) * 1.0  # non-leaf, requires grad
    render = means2d.sum() * 0 + torch.full((C, H, W, 3), 0.25) + kw["opacities"].sum() * 0
    alpha = torch.zeros(C, H, W, 1)
    info = {"means2d": means2d, "radii": torch.zeros(C, N, dtype=torch.int32)}
    TRACE["info_keys_provided"] = sorted(info)
    return render, alpha, info


_stub("ipdb")
_stub("open3d")
_stub("cv2")
_stub("plyfile", PlyData=object, PlyElement=object)
_stub("dacite", from_dict=_from_dict)
_stub("gsplat", rasterization=_recording_rasterization)
sys.path.insert(0, REF)

from edgegaussians.cameras.cameras import OpenCVCamera  # noqa: E402
from edgegaussians.data.dataparsers import EMAPDataParser  # noqa: E402
from edgegaussians.models import edge_gs  # noqa: E402
from edgegaussians.models.losses import MaskedL1Loss, WeightedL1Loss  # noqa: E402
from edgegaussians.utils import misc_utils, train_utils  # noqa: E402

CFG = json.load(open(os.path.join(REF, "configs/ABC_DexiNed.json")))


def cameras_and_edges():
    parser = EMAPDataParser(os.path.join(SCAN, "meta_data.json"))
    parser.load_views(os.path.join(SCAN, "edge_DexiNed"))
    views = parser.views
    Ks = torch.cat([v["camera"].get_K() for v in views]).numpy()
    vms = torch.cat([v["camera"].get_viewmat() for v in views]).numpy()
    np.savez_compressed(os.path.join(OUT, "cameras_00004926.npz"), Ks=Ks.astype(np.float32),
                        viewmats=vms.astype(np.float32),
                        height=views[0]["camera"].height, width=views[0]["camera"].width)
    # sparse edge maps (DexiNed, uint8) for 4 views: flat index + value of the non-zero pixels
    keep = [0, 7, 23, 41]
    sp = {}
    for k in keep:
        im = views[k]["image"].numpy().astype(np.uint8)
        nz = np.flatnonzero(im)
        sp[f"idx_{k}"] = nz.astype(np.int32)
        sp[f"val_{k}"] = im.reshape(-1)[nz]
    np.savez_compressed(os.path.join(OUT, "edges_00004926.npz"), views=np.array(keep), **sp)
    return views, keep


def quats():
    torch.manual_seed(11)
    q = misc_utils.random_quat_tensor(64) * (0.5 + torch.rand(64, 1))
    R = misc_utils.quats_to_rotmats_tensor(q)
    np.savez(os.path.join(OUT, "quats.npz"), quats=q.numpy(), rotmats=R.numpy())


def lr_table():
    model = types.SimpleNamespace(gauss_params={
        k: torch.nn.Parameter(torch.zeros(4, d))
        for k, d in (("means", 3), ("scales", 3), ("quats", 4), ("opacities", 1))})
    opts, scheds = train_utils.get_optimizers_schedulers(model, CFG["training"]["optim"])
    names = ["means", "scales", "quats", "opacities"]
    rows = []
    for epoch in range(70):
        rows.append([opts[n].param_groups[0]["lr"] for n in names])  # lr used DURING this epoch
        for n in names:
            opts[n].step()
            scheds[n].step()
    np.savez(os.path.join(OUT, "lr_table.npz"), names=np.array(names), lr=np.array(rows, dtype=np.float64))
    # the optimiser section of the config the table was produced from (values only)
    json.dump(CFG["training"]["optim"], open(os.path.join(OUT, "abc_optim_config.json"), "w"), indent=1)


def _make_model(views, n=96, seed=5):
    torch.manual_seed(seed)
    m = edge_gs.EdgeGaussianSplatting(device="cpu")
    pts = 1.1 * torch.rand(n, 3) - 0.55 + 0.5
    m.poplutate_params(seed_points=pts, viewcams=views, config=CFG["model"])
    return m


def losses_and_masks(views, keep):
    cams = [views[k]["camera"] for k in keep]
    m = _make_model(cams)
    gts = [views[k]["image"] / 255.0 for k in keep]
    m.compute_image_masks(gts)
    m.compute_weight_masks()
    out = {}
    for vi in range(len(keep)):
        em = m.edge_masks[vi]
        out[f"n_edge_{vi}"] = int(em.sum())
        wm = m.weight_masks[vi]
        out[f"w_edge_{vi}"] = float(wm[em][0])
        out[f"w_bg_{vi}"] = float(wm[~em][0])
    # losses on a 160x160 crop containing edges, view 0
    em = m.edge_masks[0]
    ys, xs = torch.where(em)
    y0 = int(ys.float().mean()) - 80
    x0 = int(xs.float().mean()) - 80
    crop = (slice(y0, y0 + 160), slice(x0, x0 + 160))
    gt = gts[0][crop].contiguous()
    torch.manual_seed(3)
    pred = torch.clamp(gt * 0.6 + 0.15 * torch.rand_like(gt), 0, 1)
    m2 = _make_model(cams[:1])
    m2.compute_image_masks([gt])
    m2.compute_weight_masks()
    out["crop"] = np.array([y0, x0, 160, 160])
    out["pred"] = pred.numpy()
    out["loss_whole"] = float(m2.compute_projection_loss(pred, gt, 0, "whole"))
    out["loss_weighted"] = float(m2.compute_projection_loss(pred, gt, 0, "weighted"))
    # bg_edge_ratio draws torch.randperm on the default CPU generator: record it
    real_randperm = torch.randperm
    rec = {}

    def _rp(n, *a, **k):
        r = real_randperm(n, *a, **k)
        rec["n"], rec["perm"] = n, r.clone()
        return r
    torch.randperm = _rp
    try:
        torch.manual_seed(17)
        out["loss_bg_edge_ratio"] = float(
            m2.compute_projection_loss(pred, gt, 0, "bg_edge_ratio", bg_edge_pixel_ratio=1.5))
    finally:
        torch.randperm = real_randperm
    out["randperm_n"] = rec["n"]
    nsel = int(1.5 * m2.edge_masks[0].sum())
    out["randperm_head"] = rec["perm"][:nsel].numpy().astype(np.int32)
    out["randperm_seed"] = 17
    # the two L1 flavours directly (losses.py:5-11)
    out["masked_l1"] = float(MaskedL1Loss()(pred, gt, m2.edge_masks[0]))
    out["weighted_l1"] = float(WeightedL1Loss()(pred, gt, m2.weight_masks[0]))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    return m, cams


def densify_cull(m, cams):
    """dup (edge_gs.py:544-576,460-474,431-457), opacity cull (:477-488,412-429,384-409),
    not-projecting cull (:578-601) on a seeded 96-Gaussian model with live Adam state."""
    opts, _ = train_utils.get_optimizers_schedulers(m, CFG["training"]["optim"])
    torch.manual_seed(23)
    for name, p in m.gauss_params.items():
        p.grad = torch.randn_like(p) * 0.01
    for o in opts.values():
        o.step()
        o.zero_grad()
    with torch.no_grad():
        m.gauss_params["opacities"].copy_(torch.logit(torch.rand(96, 1) * 0.3 + 0.01))
    m.absgrads = torch.rand(96) ** 3
    m.absgrads_normalize_factor = 4.0
    before = {k: v.detach().clone().numpy() for k, v in m.gauss_params.items()}
    state_before = {k: {s: opts[k].state[m.gauss_params[k]][s].clone().numpy()
                        for s in ("exp_avg", "exp_avg_sq")} for k in opts}
    out = {f"before_{k}": v for k, v in before.items()}
    out.update({f"before_{k}_{s}": v for k, d in state_before.items() for s, v in d.items()})
    out["absgrads"] = m.absgrads.numpy().copy()
    out["absgrads_factor"] = 4.0

    real_randn_like = torch.randn_like
    rec = {}

    def _rl(t, *a, **k):
        r = real_randn_like(t, *a, **k)
        rec["noise"] = r.clone()
        return r
    torch.randn_like = _rl
    try:
        m.duplicate_high_pos_gradients(opts)
    finally:
        torch.randn_like = real_randn_like
    out["dup_noise"] = rec["noise"].numpy()
    for k, v in m.gauss_params.items():
        out[f"dup_{k}"] = v.detach().numpy().copy()
        st = opts[k].state[m.gauss_params[k]]
        out[f"dup_{k}_exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"dup_{k}_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    out["dup_absgrads_len"] = int(m.absgrads.shape[0])
    out["dup_factor_after"] = float(m.absgrads_normalize_factor)

    m.absgrads = torch.arange(m.means.shape[0]).float()
    m.cull_gaussians_opacity(opts)
    for k, v in m.gauss_params.items():
        out[f"cull_{k}"] = v.detach().numpy().copy()
        st = opts[k].state[m.gauss_params[k]]
        out[f"cull_{k}_exp_avg"] = st["exp_avg"].numpy().copy()
    out["cull_absgrads"] = m.absgrads.numpy().copy()

    # not-projecting cull over the 4 fixture views
    n0 = m.means.shape[0]
    out["np_means_before"] = m.means.detach().numpy().copy()
    m.absgrads = torch.arange(n0).float()
    m.cull_gaussians_not_projecting(opts, min_projecting_fraction=0.1)
    out["np_kept_index"] = m.absgrads.numpy().astype(np.int64)  # surviving original indices
    np.savez_compressed(os.path.join(OUT, "densify_cull.npz"), **out)


def boundary_trace(views):
    cams = [views[0]["camera"]]
    m = _make_model(cams, n=2500, seed=1)
    torch.Tensor.cuda = lambda self, *a, **k: self  # edge_gs.py:247 hard-codes .cuda()
    m.train()
    out = m(0)
    TRACE["forward_returns"] = {k: (list(v.shape) if isinstance(v, torch.Tensor) else None)
                                for k, v in out.items()}
    img = out["rgb"][:, :, 0]
    loss = img.mean() + m.xys.sum() * 0
    loss.backward()
    m.xys.absgrad = torch.ones_like(m.xys)  # the attribute update_absgrads reads (edge_gs.py:612)
    m.update_absgrads()
    TRACE["reads_back"] = {
        "info['means2d']": "retain_grad() then .absgrad[0].norm(dim=-1) (edge_gs.py:270-275,612)",
        "info['radii'][0]": list(m.radii.shape),
        "absgrads_after_one_step": float(m.absgrads[0]),
        "absgrads_normalize_factor": float(m.absgrads_normalize_factor),
        "step": int(m.step),
    }
    json.dump(TRACE, open(os.path.join(OUT, "boundary_trace.json"), "w"), indent=1, sort_keys=True)


def filter_projection(views, keep):
    """edge_extraction/filtering.py:80-123 run as-is on seeded points and the four fixture views."""
    from edgegaussians.edge_extraction import filtering
    rng = np.random.default_rng(3)
    means = (1.1 * rng.random((3000, 3)) - 0.05).astype(np.float32)
    cams, imgs = [], []
    for k in keep:
        cam = views[k]["camera"]
        vm = cam.viewmat.cpu().numpy()
        cams.append({"K": cam.get_K().cpu().numpy()[0], "R": vm[:3, :3], "t": vm[:3, 3:], "h": cam.height,
                     "w": cam.width})
        imgs.append(views[k]["image"] / 255.0)
    out = {}
    for thr in (0.1, 0.3):
        out[f"inliers_{thr}"] = filtering.filter_by_projection(means, imgs, cams, visib_thresh=thr)
    np.savez_compressed(os.path.join(OUT, "filter_projection.npz"), means=means, views=np.array(keep), **out)


def regularizers(views):
    """update_nearest_neighbors / compute_direction_loss / compute_ratio_loss (edge_gs.py:326-380) run
    as-is (sklearn KD-tree, autograd) on a seeded 2400-Gaussian state that looks like a trained one:
    means along a few line segments plus noise, anisotropic scales, random quaternions.  Three settings:
    the shipped one (enforce_full, k = 5), Replica's k = 10, and 'enforce_half' with k = 5 and k = 10."""
    cams = [views[0]["camera"]]
    n = 2400
    g = torch.Generator().manual_seed(21)
    t = torch.rand(n, 1, generator=g)
    seg = torch.randint(0, 8, (n,), generator=g)
    a, b = torch.rand(8, 3, generator=g), torch.rand(8, 3, generator=g)
    pts = a[seg] * (1 - t) + b[seg] * t + 0.004 * torch.randn(n, 3, generator=g)
    pts[: n // 6] = torch.rand(n // 6, 3, generator=g)  # plus a uniform background
    out = {"means": pts.numpy().astype(np.float32)}
    m = _make_model(cams, n=n, seed=22)
    with torch.no_grad():
        m.gauss_params["means"].copy_(pts)
        m.gauss_params["scales"].copy_(torch.log(0.004 * (1 + 4 * torch.rand(n, 3, generator=g))))
        m.gauss_params["quats"].mul_(0.5 + torch.rand(n, 1, generator=g))  # un-normalised, like after training
    out["quats"] = m.quats.detach().numpy().copy()
    out["log_scales"] = m.scales.detach().numpy().copy()
    for method, k in (("enforce_full", 5), ("enforce_full", 10), ("enforce_half", 5), ("enforce_half", 10)):
        m.dir_loss_num_nn, m.dir_loss_enforce_method = k, method
        m.update_nearest_neighbors()
        tag = f"{method}_{k}"
        out[f"nn_{tag}"] = np.asarray(m.nn_indices).astype(np.int32)
        for p in m.gauss_params.values():
            p.grad = None
        loss = m.compute_direction_loss()
        loss.backward()
        out[f"dir_loss_{tag}"] = np.float64(loss.item())
        out[f"dir_gmeans_{tag}"] = m.means.grad.numpy().copy()
        out[f"dir_gquats_{tag}"] = m.quats.grad.numpy().copy()
        assert m.scales.grad is None or float(m.scales.grad.abs().max()) == 0.0  # argmax: no gradient to the scales
    for p in m.gauss_params.values():
        p.grad = None
    r = m.compute_ratio_loss()
    r.backward()
    out["ratio_loss"] = np.float64(r.item())
    out["ratio_gscales"] = m.scales.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "regularizers.npz"), **out)


def train_fixture(views):
    """A small end-to-end training problem cut from the reference's own data: 16 of the 50 DexiNed
    edge maps of scan 00004926 at half resolution (2x2 block mean, sparse), their cameras, and 4000 of
    the ground-truth edge points the reference's eval.py scores against (groundtruth/sampled_pts)."""
    sel = list(range(0, 48, 3))
    H, W = views[0]["camera"].height, views[0]["camera"].width
    out = {"views": np.array(sel), "height": H // 2, "width": W // 2}
    Ks, vms = [], []
    for k in sel:
        cam = views[k]["camera"]
        K = cam.get_K().cpu().numpy()[0].astype(np.float64).copy()
        K[0, 0] *= 0.5; K[1, 1] *= 0.5                      # pixel centres at integer + 0.5 in both grids
        K[0, 2] = (K[0, 2] + 0.5) * 0.5 - 0.5; K[1, 2] = (K[1, 2] + 0.5) * 0.5 - 0.5
        Ks.append(K.astype(np.float32)); vms.append(cam.viewmat.cpu().numpy().astype(np.float32))
        im = views[k]["image"].numpy().astype(np.float32).reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
        im = np.round(im).astype(np.uint8)
        nz = np.flatnonzero(im)
        out[f"idx_{k}"] = nz.astype(np.int32)
        out[f"val_{k}"] = im.reshape(-1)[nz]
    out["Ks"], out["viewmats"] = np.stack(Ks), np.stack(vms)
    # binary little-endian PLY written by Open3D: double x y z + uchar r g b per vertex
    raw = open(os.path.join(REF, "data/ABC-NEF_Edge/groundtruth/sampled_pts/00004926_0.005.ply"), "rb").read()
    head_end = raw.index(b"end_header\n") + len(b"end_header\n")
    n = int([ln for ln in raw[:head_end].decode().splitlines() if ln.startswith("element vertex")][0].split()[-1])
    rec = np.frombuffer(raw, dtype=np.dtype([("p", "<f8", 3), ("c", "u1", 3)]), count=n, offset=head_end)
    pick = np.random.default_rng(0).choice(n, 4000, replace=False)
    out["gt_points"] = rec["p"][np.sort(pick)].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "abc_00004926_train.npz"), **out)


def train_config():
    """The schedule the training loop consumes (numbers only), as parsed from configs/ABC_DexiNed.json."""
    used_model = ["if_duplicate_high_pos_grad", "dup_threshold_type", "dup_threshold_value", "dup_factor",
                  "dup_high_pos_grads_at_epoch", "if_cull_low_opacity", "cull_opacity_type", "cull_opacity_value",
                  "cull_opacity_at_epoch", "if_cull_gaussians_not_projecting", "cull_gaussians_not_projecting_at_epoch",
                  "cull_gaussians_not_projecting_threshold", "init_min_num_gaussians", "random_init_box_center",
                  "random_init_box_size", "init_dup_rand_noise_scale", "init_scales_val", "init_opacity_val"]
    tr = CFG["training"]
    out = {"model": {k: CFG["model"][k] for k in used_model},
           "training": {"num_epochs": tr["num_epochs"], "optim": tr["optim"], "loss": tr["loss"]}}
    json.dump(out, open(os.path.join(OUT, "abc_train_config.json"), "w"), indent=1, sort_keys=True)


def replica_calendar():
    """The reference's own epoch driver (train_gaussians.py:144-222) run on configs/Replica.json with a RECORDING
    stand-in for the model (its config is the reference's dataclass, filled the way dacite fills it) and a no-op
    train_epoch: the trace is which model methods the reference calls after which epoch -- in particular that a
    `cull_wayward` epoch (whose mask is never applied, edge_gs.py:498-542) still ends in `reset_absgrads`."""
    _stub("torch.utils.tensorboard", SummaryWriter=lambda *a, **k: types.SimpleNamespace(add_scalar=lambda *a, **k: None))
    _stub("edgegaussians.vis.vis_utils")
    import importlib
    tg = importlib.import_module("train_gaussians")
    cfg = json.load(open(os.path.join(REF, "configs/Replica.json")))
    mc = _from_dict(edge_gs.EdgeGaussianSplattingConfig, cfg["model"])
    trace = []

    class Rec:
        config = mc
        gauss_params = {"means": torch.zeros(5, 3)}
        epoch = -1

        def __getattr__(self, name):
            def f(*a, **k):
                trace.append([self.epoch, name])
            return f

    model = Rec()

    def fake_epoch(model_, dl, opt, dev, sw, epoch, *a, **k):
        model_.epoch = epoch
        return 0.0

    tg.train_epoch = fake_epoch
    tg.train_utils.get_optimizers_schedulers = lambda **k: ({}, {})
    tg.train_utils.save_model = lambda *a, **k: None
    tg.tqdm = lambda **k: types.SimpleNamespace(__enter__=lambda s: s, __exit__=lambda s, *a: None)
    import contextlib

    class _Bar(contextlib.AbstractContextManager):
        def set_postfix(self, *a, **k): pass
        def update(self, *a, **k): pass
        def __exit__(self, *a): return False
    tg.tqdm = lambda **k: _Bar()
    tr_cfg = cfg["training"]
    tg.train(model, tr_cfg, None, "/tmp/_eg_log", "/tmp/_eg_out", "cpu")
    used = ["if_duplicate_high_pos_grad", "dup_threshold_type", "dup_threshold_value", "dup_factor",
            "dup_high_pos_grads_at_epoch", "if_cull_low_opacity", "cull_opacity_type", "cull_opacity_value",
            "cull_opacity_at_epoch", "if_cull_gaussians_not_projecting", "cull_gaussians_not_projecting_at_epoch",
            "cull_gaussians_not_projecting_threshold", "if_cull_wayward", "cull_wayward_at_epoch",
            "init_dup_rand_noise_scale", "reset_opacity_value"]
    out = {"model": {k: getattr(mc, k) for k in used}, "if_reset_opacity_as_parsed": bool(mc.if_reset_opacity),
           "training": {"num_epochs": tr_cfg["num_epochs"], "optim": tr_cfg["optim"], "loss": tr_cfg["loss"]},
           "calls_after_epoch": [t for t in trace if t[0] >= 0]}
    json.dump(out, open(os.path.join(OUT, "replica_calendar.json"), "w"), indent=None, sort_keys=True)


if __name__ == "__main__":
    if "--only-calendar" in sys.argv:
        replica_calendar()
        raise SystemExit(0)
    views, keep = cameras_and_edges()
    if "--only-train" in sys.argv:
        train_fixture(views)
        train_config()
        raise SystemExit(0)
    if "--only-filter" in sys.argv:
        filter_projection(views, keep)
        raise SystemExit(0)
    if "--only-regularizers" in sys.argv:
        regularizers(views)
        raise SystemExit(0)
    filter_projection(views, keep)
    regularizers(views)
    train_fixture(views)
    train_config()
    quats()
    lr_table()
    m, cams = losses_and_masks(views, keep)
    densify_cull(m, cams)
    boundary_trace(views)
    replica_calendar()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
