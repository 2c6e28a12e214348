"""The oracle pinned against closed-form known answers, float64 gradcheck and algebraic invariants.

gsplat (where the rasterizer arithmetic of the reference lives) is not available, so every
constant that SURVEY.md 2.3 / 8a lists is frozen here by a case that would fail if it were
mis-remembered: +0.5 pixel centres, eps2d = 0.3 blur + compensation, the 0.999 alpha cap, the
1/255 cut, the stop BEFORE T <= 1e-4, max(0.01, b^2 - det) in the radius, 3-sigma ceil radius,
near-plane cull, the 1.3x fov clamp, row-major tile emission, (tile << 32 | depth bits) keys and
stable ordering.
"""
import math

import numpy as np
import pytest
import torch

from oracle import ref_torch as O

W, H = 64, 48
FX = FY = 100.0


def _cam(dtype=torch.float32, cxy=(W / 2, H / 2)):
    K = torch.tensor([[FX, 0, cxy[0]], [0, FY, cxy[1]], [0, 0, 1]], dtype=dtype)
    return torch.eye(4, dtype=dtype), K


def _mean_at(px, py, z, dtype=torch.float32, cxy=(W / 2, H / 2)):
    return torch.tensor([(px - cxy[0]) / FX * z, (py - cxy[1]) / FY * z, z], dtype=dtype)


def _render(means, scales, opac, quats=None, colors=None, mode="antialiased", dtype=torch.float32,
            cxy=(W / 2, H / 2), **kw):
    """cxy: principal point.  Putting it ON the pixel centre under test makes the Gaussian on-axis,
    where the perspective Jacobian is diagonal and the closed forms below hold exactly."""
    n = means.shape[0]
    vm, K = _cam(dtype, cxy)
    quats = torch.tensor([[1.0, 0, 0, 0]], dtype=dtype).repeat(n, 1) if quats is None else quats
    colors = torch.ones(n, 3, dtype=dtype) if colors is None else colors
    return O.rasterization(means, quats, scales, opac, colors, vm[None], K[None], W, H, packed=False,
                           rasterize_mode=mode, **kw)


def test_single_gaussian_on_pixel_centre():
    z, s, o = 2.0, 0.05, 0.6
    c = (20.5, 10.5)
    m = _mean_at(20.5, 10.5, z, cxy=c)[None]
    r, a, info = _render(m, torch.full((1, 3), s), torch.tensor([o]), cxy=c)
    v = (FX * s / z) ** 2  # isotropic 2D variance (on-axis)
    comp = math.sqrt(v * v / ((v + 0.3) ** 2))
    assert info["means2d"][0, 0].tolist() == pytest.approx([20.5, 10.5], abs=1e-5)
    assert float(info["depths"][0, 0]) == pytest.approx(z)
    assert float(info["conics"][0, 0, 0]) == pytest.approx(1 / (v + 0.3), rel=1e-5)
    assert float(info["conics"][0, 0, 1]) == pytest.approx(0.0, abs=1e-7)
    assert float(info["opacities"][0, 0]) == pytest.approx(o * comp, rel=1e-5)
    # pixel (i=10, j=20) has its centre exactly on the mean: sigma = 0
    assert float(r[0, 10, 20, 0]) == pytest.approx(o * comp, rel=1e-5)
    assert float(a[0, 10, 20, 0]) == pytest.approx(o * comp, rel=1e-5)
    # one pixel to the right: sigma = 0.5 / (v + 0.3)
    assert float(r[0, 10, 21, 0]) == pytest.approx(o * comp * math.exp(-0.5 / (v + 0.3)), rel=1e-5)
    # radius = ceil(3 sqrt(lambda_max)), lambda_max = v + 0.3 (+ sqrt(0.01) from the max(0.01, .) guard)
    assert int(info["radii"][0, 0]) == math.ceil(3 * math.sqrt(v + 0.3 + 0.1))
    # classic mode: no compensation
    r2, _, info2 = _render(m, torch.full((1, 3), s), torch.tensor([o]), mode="classic", cxy=c)
    assert float(r2[0, 10, 20, 0]) == pytest.approx(o, rel=1e-6)


def test_alpha_cap_cut_and_early_stop():
    z, s = 2.0, 0.2  # v = 100 -> comp ~ 0.997
    c = (8.5, 8.5)
    m = _mean_at(8.5, 8.5, z, cxy=c)[None]
    # cap: opacity 1.0 / comp -> alpha would be 1.0, capped at 0.999
    v = (FX * s / z) ** 2
    comp = v / (v + 0.3)
    r, a, _ = _render(m, torch.full((1, 3), s), torch.tensor([1.0 / comp]), cxy=c)
    assert float(a[0, 8, 8, 0]) == pytest.approx(0.999, abs=1e-6)
    # cut: alpha just below 1/255 at the centre contributes nowhere
    r, a, _ = _render(m, torch.full((1, 3), s), torch.tensor([0.999 / 255 / comp]), cxy=c)
    assert float(a.abs().max()) == 0.0
    r, a, _ = _render(m, torch.full((1, 3), s), torch.tensor([1.001 / 255 / comp]), cxy=c)
    assert float(a[0, 8, 8, 0]) == pytest.approx(1.001 / 255, rel=1e-5)
    # stop BEFORE the Gaussian that takes T to <= 1e-4: three stacked alpha = 0.999 layers
    ms = torch.stack([_mean_at(8.5, 8.5, zz, cxy=c) for zz in (2.0, 2.5, 3.0)])
    sc = torch.stack([torch.full((3,), 0.2 * zz / 2.0) for zz in (2.0, 2.5, 3.0)])
    r, a, info = _render(ms, sc, torch.full((3,), 1.0 / comp), cxy=c)
    # first: T = 1e-3; second would give 1e-6 <= 1e-4 -> the walk ends with ONE contributor
    assert float(a[0, 8, 8, 0]) == pytest.approx(0.999, abs=1e-6)
    assert float(r[0, 8, 8, 0]) == pytest.approx(0.999, abs=1e-6)
    first = int(np.nonzero(info["flatten_ids"].numpy() == 0)[0][0])
    assert int(info["last_ids"][0, 8, 8]) in range(0, info["flatten_ids"].shape[0])
    tile0 = info["isect_offsets"][0, 0, 0].item()
    assert int(info["last_ids"][0, 8, 8]) == tile0 + 0 and first >= 0


def test_transmittance_stop_threshold_from_both_sides():
    """T after the first layer is 1e-3; a second layer of alpha 0.95 would take it to 5e-5 <= 1e-4 and is NOT composited,
    one of alpha 0.85 takes it to 1.5e-4 > 1e-4 and IS (pins the 1e-4 against 1e-5 and against 2e-4)."""
    z, s = 2.0, 0.2
    c = (8.5, 8.5)
    v = (FX * s / z) ** 2
    comp = v / (v + 0.3)
    ms = torch.stack([_mean_at(8.5, 8.5, zz, cxy=c) for zz in (2.0, 2.5)])
    sc = torch.stack([torch.full((3,), 0.2 * zz / 2.0) for zz in (2.0, 2.5)])
    _, a, _ = _render(ms, sc, torch.tensor([1.0 / comp, 0.95 / comp]), cxy=c)
    assert float(a[0, 8, 8, 0]) == pytest.approx(0.999, abs=1e-6)
    _, a, _ = _render(ms, sc, torch.tensor([1.0 / comp, 0.85 / comp]), cxy=c)
    assert float(a[0, 8, 8, 0]) == pytest.approx(1 - 1e-3 * 0.15, abs=2e-6)


def test_radius_floor_is_observable():
    """lambda_max = b + sqrt(max(0.01, b^2 - det)): for an isotropic footprint b^2 - det = 0 and the floor adds 0.1 to
    the variance -- 3 sqrt(1.1) = 3.15 -> radius 4 where 3 sqrt(1.0) gives 3 (VERDICT r03 listed the floor as
    unobservable; it is visible in info["radii"] and in the tile box)."""
    z = 2.0
    s = z * math.sqrt(0.7) / FX  # 2-D variance 0.7 (+ 0.3 blur = 1.0)
    c = (20.5, 10.5)
    _, _, info = _render(_mean_at(20.5, 10.5, z, cxy=c)[None], torch.full((1, 3), s), torch.tensor([0.6]), cxy=c)
    assert int(info["radii"][0, 0]) == 4
    # ... and the floor is 0.01, not 0.1: variance 1.3 + 0.3 -> 3 sqrt(1.6 + 0.1) = 3.91 -> 4 (sqrt(0.1) would give 5)
    s2 = z * math.sqrt(1.3) / FX
    _, _, info = _render(_mean_at(20.5, 10.5, z, cxy=c)[None], torch.full((1, 3), s2), torch.tensor([0.6]), cxy=c)
    assert int(info["radii"][0, 0]) == 4


def test_two_stacked_gaussians_composite_in_depth_order():
    s = 0.05
    c = (30.5, 20.5)
    ms = torch.stack([_mean_at(30.5, 20.5, 3.0, cxy=c), _mean_at(30.5, 20.5, 2.0, cxy=c)])  # index 0 is BEHIND
    sc = torch.stack([torch.full((3,), s * 1.5), torch.full((3,), s)])  # same 2D footprint
    cols = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])
    o = torch.tensor([0.5, 0.25])
    r, a, info = _render(ms, sc, o, colors=cols, cxy=c)
    v = (FX * s / 2.0) ** 2
    comp = v / (v + 0.3)
    a_front, a_back = 0.25 * comp, 0.5 * comp
    px = r[0, 20, 30]
    assert float(px[1]) == pytest.approx(a_front, rel=1e-5)  # green (front) unattenuated
    assert float(px[0]) == pytest.approx(a_back * (1 - a_front), rel=1e-5)  # red (back) behind it
    assert float(a[0, 20, 30, 0]) == pytest.approx(1 - (1 - a_front) * (1 - a_back), rel=1e-5)
    # the sorted list of that tile has the front Gaussian (id 1) first
    t = (20 // 16) * math.ceil(W / 16) + 30 // 16
    start = int(info["isect_offsets"].reshape(-1)[t])
    assert info["flatten_ids"][start:start + 2].tolist() == [1, 0]


def test_culls_near_plane_offscreen_and_radius():
    s = torch.full((4, 3), 0.05)
    ms = torch.stack([_mean_at(10.5, 10.5, 0.005),  # in front of near = 0.01
                      _mean_at(10.5, 10.5, -1.0),   # behind the camera
                      _mean_at(-200.5, 10.5, 2.0),  # far off-screen
                      _mean_at(10.5, 10.5, 2.0)])   # visible
    _, _, info = _render(ms, s, torch.full((4,), 0.5))
    assert info["radii"][0].tolist()[:3] == [0, 0, 0] and int(info["radii"][0, 3]) > 0
    assert info["tiles_per_gauss"][0].tolist()[:3] == [0, 0, 0]


def test_tile_binning_layout():
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    # centre exactly on a tile corner: radius 5 -> 2x2 tiles; radius 17 reaches a third column/row
    m2d = np.array([[16.0, 16.0], [16.0, 16.0], [63.9, 47.9]], dtype=np.float32)
    radii = np.array([5, 17, 3], dtype=np.int32)
    depths = np.array([2.0, 1.0, 3.0], dtype=np.float32)
    tpg, ids, flat = O.isect_tiles(m2d, radii, depths, 16, tw, th)
    assert tpg.tolist() == [4, 9, 1]  # second: floor(-1/16) -> clamped 0 .. ceil(33/16) = 3 in x and y
    assert len(ids) == 14
    # key = tile << 32 | float bits of depth; tile 0 holds both Gaussians, nearer first
    assert (ids[:2] >> 32).tolist() == [0, 0] and flat[:2].tolist() == [1, 0]
    assert (ids[0] & 0xffffffff) == np.float32(1.0).view(np.int32)
    offs = O.isect_offset_encode(ids, tw, th).reshape(-1)
    assert offs[0] == 0 and offs[1] == 2 and offs[-1] == 13  # last tile (bottom-right) holds the third
    # ties in depth keep Gaussian-index order (stable sort)
    tpg, ids, flat = O.isect_tiles(np.array([[8, 8], [8, 8], [8, 8]], np.float32), np.array([2, 2, 2], np.int32),
                                   np.array([1.5, 1.5, 1.5], np.float32), 16, tw, th)
    assert flat.tolist() == [0, 1, 2]


def test_fov_clamp_enters_the_jacobian():
    """A Gaussian outside the 1.3x frustum whose footprint still reaches the image: its 2-D covariance is formed with
    the CLAMPED x / z in the Jacobian, its mean2d with the true one."""
    z = 1.0
    lim = 1.3 * 0.5 * W / FX
    x = 1.5 * lim * z  # mean2d.x = 100 * 0.624 + 32 = 94.4: off-screen (W = 64)
    sg = 0.15          # ~15 px: the 3-sigma box reaches back into the image
    means = torch.tensor([[x, 0.0, z]])
    vm, K = _cam()
    radii, m2d, dep, conic, comp = O.project(means, torch.tensor([[1.0, 0, 0, 0]]), torch.full((1, 3), sg), vm, K, W, H)
    assert int(radii[0]) > 0, "the case must not be culled"
    assert float(m2d[0, 0]) == pytest.approx(FX * x / z + W / 2, rel=1e-6)  # unclamped
    tx = lim * z
    J = torch.tensor([[FX / z, 0, -FX * tx / z ** 2], [0, FY / z, 0.0]])
    cov = J @ (sg ** 2 * torch.eye(3)) @ J.T + 0.3 * torch.eye(2)
    want = torch.linalg.inv(cov)
    assert conic[0].tolist() == pytest.approx([float(want[0, 0]), float(want[0, 1]), float(want[1, 1])], rel=1e-5, abs=1e-9)
    # (with the unclamped x / z the first entry would be 1 / (225 (1 + 0.624^2) + 0.3) instead of 1 / (225 (1 + 0.416^2) + 0.3))
    assert abs(float(conic[0, 0]) - 1.0 / (sg ** 2 * FX ** 2 * (1 + (x / z) ** 2) + 0.3)) > 1e-4 * float(conic[0, 0])


def test_unit_colour_identity_and_order_independence():
    g = torch.Generator().manual_seed(0)
    n = 40
    means = torch.stack([_mean_at(5 + 50 * float(torch.rand((), generator=g)), 5 + 35 * float(torch.rand((), generator=g)),
                                  1.5 + 2 * float(torch.rand((), generator=g))) for _ in range(n)])
    scales = 0.02 + 0.05 * torch.rand(n, 3, generator=g)
    quats = torch.randn(n, 4, generator=g)
    opac = 0.05 + 0.5 * torch.rand(n, generator=g)
    r, a, info = _render(means, scales, opac, quats=quats)
    # colours == 1, no background: every channel equals the accumulated alpha
    assert torch.allclose(r[..., 0], a[..., 0], atol=2e-6) and torch.allclose(r[..., 1], r[..., 2])
    # permuting the Gaussian order changes nothing (depths are distinct)
    perm = torch.randperm(n, generator=g)
    r2, a2, _ = _render(means[perm], scales[perm], opac[perm], quats=quats[perm])
    assert torch.allclose(r, r2, atol=2e-6) and torch.allclose(a, a2, atol=2e-6)


def test_gradcheck_projection_and_full_pipeline_float64():
    g = torch.Generator().manual_seed(1)
    dt = torch.float64
    n = 5
    means = torch.stack([_mean_at(10 + 40 * float(torch.rand((), generator=g)), 10 + 25 * float(torch.rand((), generator=g)),
                                  2.0 + float(torch.rand((), generator=g)), dt) for _ in range(n)]).requires_grad_(True)
    quats = torch.randn(n, 4, generator=g, dtype=dt).requires_grad_(True)
    scales = (0.03 + 0.05 * torch.rand(n, 3, generator=g, dtype=dt)).requires_grad_(True)
    opac = (0.2 + 0.5 * torch.rand(n, generator=g, dtype=dt)).requires_grad_(True)
    vm, K = _cam(dt)

    def proj(m, q, s):
        _, m2d, dep, con, comp = O.project(m, q, s, vm, K, W, H)
        return m2d.sum() * 0.01 + (con * torch.arange(3, dtype=dt)).sum() + dep.sum() * 0.1

    assert torch.autograd.gradcheck(proj, (means, quats, scales), eps=1e-6, atol=1e-5, rtol=1e-4)

    cols = torch.rand(n, 3, generator=g, dtype=dt)
    wr = torch.rand(H, W, 3, generator=g, dtype=dt)

    def full(m, q, s, o):
        r, a, _ = O.rasterization(m, q, s, o, cols, vm[None], K[None], W, H, packed=False, rasterize_mode="classic")
        return (r[0] * wr).sum() + (a ** 2).sum()

    # classic mode: the compensation's +1e-6 guarded backward (gsplat's) is deliberately not the
    # exact derivative, everything else is
    assert torch.autograd.gradcheck(full, (means, quats, scales, opac), eps=1e-6, atol=1e-5, rtol=1e-3)


def test_absgrad_is_the_per_pixel_absolute_sum():
    g = torch.Generator().manual_seed(2)
    n = 12
    means = torch.stack([_mean_at(20 + 20 * float(torch.rand((), generator=g)), 15 + 15 * float(torch.rand((), generator=g)),
                                  2.0 + float(torch.rand((), generator=g))) for _ in range(n)]).requires_grad_(True)
    scales = torch.full((n, 3), 0.06)
    opac = torch.full((n,), 0.4)
    r, a, info = _render(means, scales, opac, absgrad=True)
    info["means2d"].retain_grad()
    w = torch.randn(H, W, generator=g)
    (r[0, ..., 0] * w).sum().backward()
    ag, gr = info["means2d"].absgrad, info["means2d"].grad
    assert ag.shape == (1, n, 2) and bool((ag >= gr.abs() - 1e-7).all()) and float(ag.sum()) > float(gr.abs().sum())
    # with a one-signed upstream gradient on a single pixel the two coincide
    means2 = means.detach().clone().requires_grad_(True)
    r, a, info = _render(means2, scales, opac, absgrad=True)
    info["means2d"].retain_grad()
    r[0, 20, 30, 0].backward()
    assert torch.allclose(info["means2d"].absgrad, info["means2d"].grad.abs(), atol=1e-9)


def test_adam_reference_matches_torch_optim():
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(50, 3, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=2e-3)
    q, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(1, 6):
        gr = torch.randn(50, 3, generator=g) * 0.01
        p.grad = gr.clone()
        opt.step()
        q, m, v = O.adam_reference(q, gr, m, v, step, 2e-3)
    assert torch.allclose(q, p.data, rtol=1e-6, atol=1e-8)
