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


def assert_close(a, b, rtol=1e-4, max_bad=0.0, name=""):
    """north_star tolerance: 1e-4 relative.  `max_bad` admits the few elements whose value hinges
    on a float-borderline branch (alpha >= 1/255, transmittance stop, radius ceil) that flips
    between two correct implementations (different exp / rounding order)."""
    fb = frac_bad(a, b, rtol)
    assert fb <= max_bad, f"{name}: {fb:.2e} of elements off by > {rtol} (allowed {max_bad}); norm-rel {rel_err(a, b):.3e}"
