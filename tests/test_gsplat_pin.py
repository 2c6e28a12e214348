"""Both oracles against gsplat 1.0.0's own reference implementation -- ONCE THE FIXTURE EXISTS.

tests/golden/gsplat_pin.npz is written by tools/pin_oracle_to_gsplat.py on a machine that holds a gsplat 1.0.0 source
tree (requirements.txt:64 of the reference; not vendored, not installable in the build image: DESIGN.md section 2,
"parity unpinned").  While the fixture is absent the comparison is SKIPPED with that reason and the rasterizer
arithmetic stays pinned only by the closed-form / mutation tests (tests/test_oracle.py, tests/test_oracle_mutations.py).
The second test exercises the pin script's plumbing against a stand-in module with gsplat's function names built from
this repository's oracle: it proves the script runs and writes the fields the comparison reads -- it pins nothing."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "gsplat_pin.npz")


def _compare(d, strict_version=True):
    from oracle import c_oracle as CO
    from oracle import ref_torch as O
    if strict_version:
        assert str(d["gsplat_version"]).startswith("1.0"), f"the reference pins gsplat==1.0.0, the fixture holds {d['gsplat_version']}"
    for i in range(int(d["n_scenes"])):
        g = lambda k: d[f"s{i}_{k}"]  # noqa: E731
        W, H = int(g("width")), int(g("height"))
        means, quats, scales, opac = (torch.from_numpy(g(k)) for k in ("means", "quats", "scales", "opacities"))
        vm, K = torch.from_numpy(g("viewmat")), torch.from_numpy(g("K"))
        # projection: the torch oracle (covariance form) and the C oracle (factor form)
        radii, m2d, dep, conic, comp = O.project(means, quats, scales, vm, K, W, H)
        fw = CO.rasterize(g("means"), g("quats"), g("scales"), g("opacities"), np.ones((means.shape[0], 3), np.float32),
                          g("viewmat"), g("K"), W, H)
        border = CO.project_borderline(g("means"), g("quats"), g("scales"), g("viewmat"), g("K"), W, H) > 0
        assert border.mean() <= 0.02
        ok = ~border
        for name, mine in (("torch oracle", radii.numpy()), ("C oracle", fw["radii"])):
            assert np.array_equal(mine[ok], g("radii")[ok]), f"{name}: radii differ from gsplat outside the borderline set"
        vis = ok & (g("radii") > 0)
        for name, a, b in (("means2d", m2d.detach().numpy(), g("means2d")), ("depths", dep.detach().numpy(), g("depths")),
                           ("conics", conic.detach().numpy(), g("conics")), ("compensations", comp.detach().numpy(), g("compensations")),
                           ("C means2d", fw["means2d"], g("means2d")), ("C conics", fw["conics"], g("conics")),
                           ("C compensations", fw["comps"], g("compensations"))):
            err = np.abs(a[vis] - b[vis]).max() / max(np.abs(b[vis]).max(), 1e-30)
            assert err <= 1e-4, (name, err)
        # binning on gsplat's OWN floats: integer work, bit-exact
        tw, th = -(-W // 16), -(-H // 16)
        tpg, ids, flat = O.isect_tiles(g("means2d"), g("radii"), g("depths"), 16, tw, th)
        assert np.array_equal(tpg, g("tiles_per_gauss")) and np.array_equal(ids, g("isect_ids"))
        assert np.array_equal(flat, g("flatten_ids"))
        assert np.array_equal(O.isect_offset_encode(ids, tw, th).reshape(-1), g("isect_offsets").reshape(-1))
        if int(d["captured_full_call"]):
            r, a, info = O.rasterization(means.requires_grad_(True), quats.requires_grad_(True), scales.requires_grad_(True),
                                         opac.requires_grad_(True), torch.ones(means.shape[0], 3), vm[None], K[None], W, H,
                                         packed=False, absgrad=True, rasterize_mode="antialiased")
            from tests.util import assert_close, borderline_pixel_mask
            pm = borderline_pixel_mask(fw).numpy()
            assert_close(r[0].detach().numpy()[~pm], g("render")[~pm], name="render vs gsplat")
            assert_close(a[0, ..., 0].detach().numpy()[~pm], g("alpha")[~pm], name="alpha vs gsplat")
            # (gradients: the fixture's cotangent is nonzero on the borderline pixels as well, so elementwise 1e-4 with
            # the few Gaussians touching them excused is what can be asserted from a fixed fixture)
            info["means2d"].retain_grad()
            (torch.from_numpy(g("cotangent")) * r[0, ..., 0]).sum().backward()
            for name, mine in (("v_means", means.grad), ("v_quats", quats.grad), ("v_scales", scales.grad), ("v_opacities", opac.grad)):
                assert_close(mine, g(name), rtol=1e-4, max_bad=5e-3, name=name)


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/gsplat_pin.npz absent: gsplat 1.0.0 is not reachable from "
                    "the build image (run tools/pin_oracle_to_gsplat.py where a checkout exists); parity stays 'unpinned'")
def test_oracles_match_gsplat_reference_implementation():
    _compare(np.load(PIN))


def test_pin_script_plumbing_against_a_stand_in(tmp_path):
    """NOT a pin: a module with gsplat 1.0.0's function names, implemented by this repository's own oracle, stands where
    gsplat/cuda/_torch_impl.py would be; the script must run, write every field, and the comparison must accept it."""
    pkg = tmp_path / "gsplat" / "cuda"
    pkg.mkdir(parents=True)
    (tmp_path / "gsplat" / "version.py").write_text('__version__ = "0.0.0-stand-in"\n')
    (pkg / "_torch_impl.py").write_text(textwrap.dedent(f'''
        import sys, numpy as np, torch
        sys.path.insert(0, {ROOT!r})
        from oracle import ref_torch as O
        _state = {{}}
        def _quat_scale_to_covar_preci(quats, scales, compute_covar=True, compute_preci=True, triu=False):
            _state["q"], _state["s"] = quats, scales
            R = O.quat_to_rotmat(quats)
            M = R * scales[:, None, :]
            return M @ M.transpose(1, 2), None
        def _fully_fused_projection(means, covars, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                                    calc_compensations=False):
            r, m2d, dep, con, comp = O.project(means, _state["q"], _state["s"], viewmats[0], Ks[0], width, height)
            return r[None], m2d[None], dep[None], con[None], comp[None]
        def _isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True):
            tpg, ids, flat = O.isect_tiles(means2d[0].detach().numpy(), radii[0].numpy(), depths[0].detach().numpy(),
                                           tile_size, tile_width, tile_height)
            return torch.from_numpy(tpg)[None], torch.from_numpy(ids), torch.from_numpy(flat)
        def _isect_offset_encode(isect_ids, C, tile_width, tile_height):
            return torch.from_numpy(O.isect_offset_encode(isect_ids.numpy(), tile_width, tile_height))[None]
    '''))
    out = tmp_path / "pin.npz"
    env = dict(os.environ, GSPLAT_SRC=str(tmp_path), GSPLAT_PIN_OUT=str(out), GSPLAT_PIN_FULL="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_oracle_to_gsplat.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    assert str(d["gsplat_version"]) == "0.0.0-stand-in" and int(d["n_scenes"]) == 3 and int(d["captured_full_call"]) == 0
    _compare(d, strict_version=False)
