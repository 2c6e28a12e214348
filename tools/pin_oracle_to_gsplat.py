#!/usr/bin/env python3
"""Pins the oracle's rasterizer arithmetic to gsplat 1.0.0 the day its source is reachable.

The reference binds its rasterizer through `from gsplat import rasterization` (edgegaussians/models/edge_gs.py:8,250-268;
requirements.txt:64 `gsplat==1.0.0`).  gsplat is neither vendored in /root/reference nor installed in the build image and
there is no network, so oracle/ref_torch.py and oracle/eg_oracle.c RESTATE its arithmetic ("parity unpinned", DESIGN.md
section 2).  This script closes the gap with ONE command on any machine that holds a gsplat 1.0.0 checkout (no CUDA needed
for the projection / binning part):

    GSPLAT_SRC=/path/to/gsplat-1.0.0 python tools/pin_oracle_to_gsplat.py            # writes tests/golden/gsplat_pin.npz
    python -m pytest tests/test_gsplat_pin.py                                          # both oracles against it

It loads gsplat/cuda/_torch_impl.py BY PATH (pure PyTorch; the package __init__ would try to build the CUDA extension),
runs its reference functions -- _quat_scale_to_covar_preci, _fully_fused_projection, _isect_tiles, _isect_offset_encode --
on seeded scenes, and, when `import gsplat` itself works on that machine (a CUDA build), also the full
`gsplat.rasterization(...)` call exactly as the reference makes it, forward AND backward.  Only DATA is written: inputs
and gsplat's outputs (no gsplat source).  The fixture records the gsplat version string and which parts were captured.
"""
from __future__ import annotations

import importlib.util
import inspect
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegaussians_amd import synth  # noqa: E402  (seeded scene generator: build code, no reference content)

SCENES = [dict(n=1200, w=112, h=80, seed=0, scale=0.02), dict(n=1200, w=112, h=80, seed=1, scale=0.02),
          dict(n=3000, w=160, h=96, seed=2, scale=0.01)]


def load_torch_impl(src: str):
    path = os.path.join(src, "gsplat", "cuda", "_torch_impl.py")
    if not os.path.exists(path):
        raise SystemExit(f"{path} not found: GSPLAT_SRC must point at a gsplat 1.0.0 source tree")
    spec = importlib.util.spec_from_file_location("_gsplat_torch_impl", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def version_of(src: str) -> str:
    v = os.path.join(src, "gsplat", "version.py")
    if os.path.exists(v):
        ns = {}
        exec(open(v).read(), ns)  # a one-line `__version__ = "1.0.0"`
        return str(ns.get("__version__", "unknown"))
    return "unknown"


def call(fn, **kw):
    """Calls fn with the keyword arguments its signature knows (minor versions renamed / added optional ones)."""
    names = inspect.signature(fn).parameters
    return fn(**{k: v for k, v in kw.items() if k in names})


def capture(impl, sc, view: int, full=None) -> dict:
    means, quats = sc.means, sc.quats
    scales, opac = torch.exp(sc.log_scales), torch.sigmoid(sc.logit_opacities).reshape(-1)
    vm, K = sc.viewmats[view:view + 1], sc.Ks[view:view + 1]
    W, H, tile = sc.width, sc.height, 16
    tw, th = -(-W // tile), -(-H // tile)
    out = dict(means=means, quats=quats, scales=scales, opacities=opac, viewmat=vm[0], K=K[0],
               width=np.int32(W), height=np.int32(H))
    covars, _ = call(impl._quat_scale_to_covar_preci, quats=quats, scales=scales, compute_covar=True, compute_preci=False,
                     triu=False)
    radii, means2d, depths, conics, comps = call(impl._fully_fused_projection, means=means, covars=covars, viewmats=vm,
                                                 Ks=K, width=W, height=H, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                                                 calc_compensations=True)
    out.update(covars=covars, radii=radii[0].to(torch.int32), means2d=means2d[0], depths=depths[0], conics=conics[0],
               compensations=comps[0])
    tpg, isect_ids, flatten_ids = call(impl._isect_tiles, means2d=means2d, radii=radii, depths=depths, tile_size=tile,
                                       tile_width=tw, tile_height=th, sort=True)
    offsets = call(impl._isect_offset_encode, isect_ids=isect_ids, C=1, n_cameras=1, tile_width=tw, tile_height=th)
    out.update(tiles_per_gauss=tpg[0].to(torch.int32), isect_ids=isect_ids.to(torch.int64),
               flatten_ids=flatten_ids.to(torch.int32), isect_offsets=offsets[0].to(torch.int32))
    if full is not None:  # the reference's own call (edge_gs.py:250-268), forward + backward, on the CUDA build
        dev = "cuda"
        p = [t.clone().to(dev).requires_grad_(True) for t in (means, quats, scales, opac)]
        render, alpha, info = full(means=p[0], quats=p[1], scales=p[2], opacities=p[3],
                                   colors=torch.ones(means.shape[0], 3, device=dev), viewmats=vm.to(dev), Ks=K.to(dev),
                                   width=W, height=H, packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        g = torch.Generator().manual_seed(7)
        wgt = torch.rand(H, W, generator=g).to(dev)
        (wgt * render[0, ..., 0]).sum().backward()
        out.update(render=render[0].detach().cpu(), alpha=alpha[0, ..., 0].detach().cpu(), cotangent=wgt.cpu(),
                   v_means=p[0].grad.cpu(), v_quats=p[1].grad.cpu(), v_scales=p[2].grad.cpu(), v_opacities=p[3].grad.cpu(),
                   absgrad=info["means2d"].absgrad[0].cpu())
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}


def main(argv):
    src = os.environ.get("GSPLAT_SRC") or (argv[1] if len(argv) > 1 else None)
    if not src:
        raise SystemExit(__doc__)
    out_path = os.environ.get("GSPLAT_PIN_OUT", os.path.join(ROOT, "tests", "golden", "gsplat_pin.npz"))
    impl = load_torch_impl(src)
    full = None
    if os.environ.get("GSPLAT_PIN_FULL", "auto") != "0" and torch.cuda.is_available():
        try:
            sys.path.insert(0, src)
            sys.modules.pop("gsplat", None)  # (this repository ships a `gsplat` shim of its own: not that one)
            import gsplat as real_gsplat
            if os.path.realpath(os.path.dirname(real_gsplat.__file__)).startswith(os.path.realpath(src)):
                full = real_gsplat.rasterization
        except Exception as e:  # noqa: BLE001
            print(f"full rasterization not captured ({e!r}): projection + binning only")
    data = {"gsplat_version": np.array(version_of(src)), "n_scenes": np.int32(len(SCENES)),
            "captured_full_call": np.int32(full is not None)}
    for i, cfg in enumerate(SCENES):
        sc = synth.make_scene(cfg["n"], 2, cfg["w"], cfg["h"], seed=cfg["seed"], spread_opacity=True, scale=cfg["scale"])
        for k, v in capture(impl, sc, 0, full).items():
            data[f"s{i}_{k}"] = v
    np.savez_compressed(out_path, **data)
    print(f"wrote {out_path}: gsplat {data['gsplat_version']}, {len(SCENES)} scenes, full call: {bool(full)}")


if __name__ == "__main__":
    main(sys.argv)
