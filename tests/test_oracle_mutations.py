"""Mutation check of the restated gsplat constants ("harden what parity can still be hardened" (d)).

gsplat 1.0.0 is not in this image, so the constants of the rasterizer arithmetic are restated from SURVEY.md 2.3 / 8a
(oracle/ref_torch.py header).  A known-answer test only pins a constant if it FAILS when the constant is wrong: here
every constant is flipped to a plausible mis-remembering and at least one closed-form test of tests/test_oracle.py must
fail.  (The plain-C oracle carries the same constants; it is held to the torch oracle on every pixel and every
gradient element by tests/test_c_oracle.py, so a constant pinned here is pinned there.)"""
import pytest

from oracle import ref_torch as O
from tests import test_oracle as KA

KNOWN_ANSWER_TESTS = [KA.test_single_gaussian_on_pixel_centre, KA.test_alpha_cap_cut_and_early_stop,
                      KA.test_transmittance_stop_threshold_from_both_sides, KA.test_radius_floor_is_observable,
                      KA.test_two_stacked_gaussians_composite_in_depth_order,
                      KA.test_culls_near_plane_offscreen_and_radius, KA.test_fov_clamp_enters_the_jacobian]

MUTATIONS = [("ALPHA_MAX", 0.99), ("ALPHA_MAX", 0.9999), ("ALPHA_MAX", 1.0),
             ("ALPHA_MIN", 1.0 / 256.0), ("ALPHA_MIN", 0.004), ("ALPHA_MIN", 0.0),
             ("T_STOP", 1e-3), ("T_STOP", 1e-5), ("T_STOP", 2e-4),
             ("STOP_BEFORE", False),
             ("EPS2D", 0.25), ("EPS2D", 0.35), ("EPS2D", 0.0),
             ("FOV_CLAMP", 1.2), ("FOV_CLAMP", 1.5), ("FOV_CLAMP", 1e9),
             ("PIXEL_CENTRE", 0.0), ("PIXEL_CENTRE", 1.0),
             ("RADIUS_SIGMAS", 2.5), ("RADIUS_SIGMAS", 3.5),
             ("RADIUS_DET_FLOOR", 0.0), ("RADIUS_DET_FLOOR", 0.1)]


def _failures():
    failed = []
    for t in KNOWN_ANSWER_TESTS:
        try:
            t()
        except AssertionError:
            failed.append(t.__name__)
    return failed


def test_unmutated_oracle_passes_every_known_answer_test():
    assert _failures() == []


@pytest.mark.parametrize("name,value", MUTATIONS, ids=[f"{n}={v}" for n, v in MUTATIONS])
def test_a_wrong_constant_fails_a_known_answer_test(monkeypatch, name, value):
    assert hasattr(O, name)
    monkeypatch.setattr(O, name, value)
    if name == "EPS2D":  # (the blur is also a default argument of the entry points: the reference passes none)
        for fn in (O.project, O.rasterization):
            d = list(fn.__defaults__)
            kw = fn.__kwdefaults__
            if kw and "eps2d" in kw:
                monkeypatch.setitem(kw, "eps2d", value)
            else:
                import inspect
                names = [p for p in inspect.signature(fn).parameters.values() if p.default is not inspect.Parameter.empty]
                idx = [i for i, p in enumerate(names) if p.name == "eps2d"][0]
                d[idx] = value
                monkeypatch.setattr(fn, "__defaults__", tuple(d))
    failed = _failures()
    assert failed, f"{name} = {value} passes every known-answer test: the constant is not pinned"
