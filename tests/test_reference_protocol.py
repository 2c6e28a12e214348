"""The reference's OWN model class driven through the operator surface this repo implements.

Runs only in the build container (it imports /root/reference; skipped on the GPU box, where that
tree does not exist).  `gsplat.rasterization` is bound to the CPU ORACLE, whose signature, returns
and `info` contract are the same as the HIP drop-in's (tests/test_gpu_parity.py checks the two
against each other on the GPU), so this test proves the boundary protocol against the real caller:
forward -> compute_projection_loss -> backward -> update_absgrads -> 4x Adam, and the densify path
that consumes absgrads (edge_gs.py:197-324, 544-613; train_gaussians.py:81-106)."""
import dataclasses
import json
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture()
def ref_modules(monkeypatch):
    from oracle import ref_torch as O

    def from_dict(data_class, data):
        names = {f.name for f in dataclasses.fields(data_class)}
        return data_class(**{k: v for k, v in data.items() if k in names})
    saved = dict(sys.modules)
    for name, attrs in (("ipdb", {}), ("open3d", {}), ("plyfile", {"PlyData": object, "PlyElement": object}),
                        ("dacite", {"from_dict": from_dict}), ("gsplat", {"rasterization": O.rasterization})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        monkeypatch.setitem(sys.modules, name, m)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)  # edge_gs.py:247 hard-codes .cuda()
    from edgegaussians.cameras.cameras import OpenCVCamera
    from edgegaussians.models import edge_gs
    from edgegaussians.utils import train_utils
    yield edge_gs, train_utils, OpenCVCamera
    for k in list(sys.modules):
        if k.startswith("edgegaussians") or k in ("gsplat", "dacite", "ipdb", "open3d", "plyfile"):
            if k not in saved:
                del sys.modules[k]


def test_reference_model_class_runs_a_training_step(ref_modules):
    edge_gs, train_utils, OpenCVCamera = ref_modules
    from edgegaussians_amd import synth
    cfg = json.load(open(os.path.join(REF, "configs/ABC_DexiNed.json")))
    W = H = 96
    sc = synth.make_scene(400, 2, W, H, seed=3, scale=0.02)
    cams = []
    for v in range(2):
        K = sc.Ks[v].numpy()
        cam = OpenCVCamera(height=H, width=W, K=K, R=sc.viewmats[v, :3, :3].numpy(), t=sc.viewmats[v, :3, 3].numpy())
        cams.append(cam)
    model = edge_gs.EdgeGaussianSplatting(device="cpu")
    model.poplutate_params(seed_points=sc.means.clone(), viewcams=cams, config=cfg["model"])
    model.compute_image_masks([sc.gt[0], sc.gt[1]])
    model.compute_weight_masks()
    opts, scheds = train_utils.get_optimizers_schedulers(model, cfg["training"]["optim"])
    model.train()
    before = model.means.detach().clone()
    for step, (idx, strategy) in enumerate([(0, "whole"), (1, "weighted"), (0, "bg_edge_ratio")]):
        out = model(idx)                                                  # edge_gs.py:617-623 -> :250-268
        assert out["rgb"].shape == (H, W, 3) and out["accumulation"].shape == (H, W, 1)
        img = out["rgb"][:, :, 0]
        loss = model.compute_projection_loss(img, sc.gt[idx], image_index=idx, strategy=strategy,
                                             bg_edge_pixel_ratio=1.0)
        loss.backward()
        model.update_absgrads()                                           # reads means2d.absgrad (:612)
        for o in opts.values():
            o.step()
            o.zero_grad()
    assert model.absgrads.shape == (400,) and float(model.absgrads.max()) > 0
    assert model.absgrads_normalize_factor == 4 and model.step == 3
    assert float((model.means.detach() - before).abs().max()) > 0
    assert model.radii.dtype == torch.int32 and model.xys.shape == (1, 400, 2)
    # the densify event that consumes the accumulated absgrads
    n0 = model.means.shape[0]
    model.duplicate_high_pos_gradients(opts)
    assert model.means.shape[0] >= n0 and model.absgrads.shape[0] == model.means.shape[0]
