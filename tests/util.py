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
