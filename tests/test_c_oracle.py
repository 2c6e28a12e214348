"""The two CPU oracles against each other: oracle/eg_oracle.c (plain C, per-pixel sequential walk,
hand-derived backward with transmittance recovery) vs oracle/ref_torch.py (dense tensors, autograd).
They share no code, so agreement pins the restated semantics from two independent derivations."""
import numpy as np
import pytest
import torch

from edgegaussians_amd import synth
from oracle import c_oracle as CO
from oracle import ref_torch as O
from tests.util import assert_close, borderline_pixel_mask, clean_scene, record_cpu, rel_err


@pytest.fixture(scope="module")
def built():
    CO.build()
    return CO


def _scene(n=1200, w=112, h=80, seed=0, scale=0.02):
    return synth.make_scene(n, 2, w, h, seed=seed, spread_opacity=True, scale=scale)


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("mode", ["antialiased", "classic"])
def test_forward_matches_torch_oracle(built, mode, seed):
    """STRICT form (round 4; round 3 admitted 0.2 % radius mismatches and 0.2 % of the pixels): the Gaussians whose
    integer decisions lie within 2e-5 of a float threshold are taken out of the scene (tests/util.py: clean_scene),
    then radii are EQUAL for every Gaussian and render / alpha agree to 1e-4 on EVERY pixel outside the quantified
    borderline pixel set -- max_bad = 0 -- between two implementations that share no code and no formulation (the
    torch oracle projects through the 3-D covariance, the C one through the factor P = J W R S)."""
    sc, removed = clean_scene(_scene(seed=seed), [0])
    assert removed <= 0.02 * 1200
    scales, opac = torch.exp(sc.log_scales), torch.sigmoid(sc.logit_opacities).squeeze(-1)
    colors = torch.rand(sc.means.shape[0], 3, generator=torch.Generator().manual_seed(1))
    r, a, info = O.rasterization(sc.means, sc.quats, scales, opac, colors, sc.viewmats[:1], sc.Ks[:1], sc.width,
                                 sc.height, packed=False, rasterize_mode=mode)
    fw = built.rasterize(sc.means.numpy(), sc.quats.numpy(), scales.numpy(), opac.numpy(), colors.numpy(),
                         sc.viewmats[0].numpy(), sc.Ks[0].numpy(), sc.width, sc.height, antialiased=(mode == "antialiased"))
    ro = info["radii"][0].numpy()
    assert np.array_equal(fw["radii"], ro), f"{int((fw['radii'] != ro).sum())} radii differ on a cleaned scene"
    vis = ro > 0
    assert_close(fw["means2d"][vis], info["means2d"][0].detach().numpy()[vis], name="means2d")
    assert_close(fw["conics"][vis], info["conics"][0].detach().numpy()[vis], name="conics")
    if mode == "antialiased":
        assert_close(fw["comps"][vis], (info["opacities"][0] / opac.clamp_min(1e-12)).numpy()[vis], name="comps")
    border = borderline_pixel_mask(fw).numpy()
    assert border.mean() <= 0.03
    ok = ~border
    assert_close(fw["render"][ok], r[0].detach().numpy()[ok], name="render")   # max_bad = 0
    assert_close(fw["alphas"][ok], a[0, ..., 0].detach().numpy()[ok], name="alpha")
    # inside the borderline set the outcome is BOUNDED: a pixel differs from the other implementation's by at most
    # what the Gaussians sitting on a threshold can contribute -- the two renders are both within [0, 1] accumulations of
    # the same list, so the difference is below the largest single contribution, alpha_max * T <= 0.999
    d_in = np.abs(fw["alphas"][border] - a[0, ..., 0].detach().numpy()[border])
    record_cpu("c_vs_torch_oracle_forward", mode=mode, seed=seed, removed=removed, borderline_pixels=int(border.sum()),
               max_err_outside=rel_err(fw["render"][ok], r[0].detach().numpy()[ok]),
               max_alpha_diff_inside=float(d_in.max()) if d_in.size else 0.0)
    # integer pipeline on identical floats: feed the torch oracle's floats to the C binning
    m2d, dep = info["means2d"][0].detach().numpy(), info["depths"][0].detach().numpy()
    lib = built.load()
    import ctypes as C
    N = m2d.shape[0]
    tpg = np.zeros(N, np.int32)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    m2d, dep, ro = np.ascontiguousarray(m2d), np.ascontiguousarray(dep), np.ascontiguousarray(ro)
    M = int(lib.ego_isect_count(ptr(m2d), ptr(ro), N, sc.width, sc.height, ptr(tpg)))
    ids, flat = np.zeros(max(M, 1), np.int64), np.zeros(max(M, 1), np.int32)
    offs = np.zeros(info["isect_offsets"].numel(), np.int32)
    lib.ego_isect_emit_sort(ptr(m2d), ptr(ro), ptr(dep), N, sc.width, sc.height, C.c_int64(M), ptr(ids), ptr(flat), ptr(offs))
    assert np.array_equal(tpg, info["tiles_per_gauss"][0].numpy())
    assert np.array_equal(ids[:M], info["isect_ids"].numpy()) and np.array_equal(flat[:M], info["flatten_ids"].numpy())
    assert np.array_equal(offs, info["isect_offsets"].numpy().reshape(-1))


@pytest.mark.parametrize("seed", [3, 4])
def test_backward_general_colours_matches_autograd(built, seed):
    """The C oracle's hand-derived backward (composite VJP with transmittance recovery + projection VJP) against autograd
    through the dense torch oracle, general colours and an alpha cotangent.  STRICT: cleaned scene, zero cotangent on the
    borderline pixels, then 1e-4 on EVERY element of every gradient (round 3: 2e-4 with 0.5 % of the elements excused)."""
    sc, removed = clean_scene(_scene(n=900, w=96, h=64, seed=seed), [0])
    N = sc.means.shape[0]
    g = torch.Generator().manual_seed(2)
    colors0 = torch.rand(N, 3, generator=g)
    wr, wa = torch.rand(sc.height, sc.width, 3, generator=g), torch.rand(sc.height, sc.width, generator=g) * 0.1
    p = [t.clone().requires_grad_(True) for t in (sc.means, sc.quats, torch.exp(sc.log_scales),
                                                  torch.sigmoid(sc.logit_opacities).squeeze(-1), colors0)]
    fw = built.rasterize(*[t.detach().numpy() for t in p], sc.viewmats[0].numpy(), sc.Ks[0].numpy(), sc.width, sc.height)
    border = borderline_pixel_mask(fw)
    assert float(border.float().mean()) <= 0.03
    wr[border] = 0.0
    wa[border] = 0.0
    r, a, info = O.rasterization(*p, sc.viewmats[:1], sc.Ks[:1], sc.width, sc.height, packed=False, absgrad=True,
                                 rasterize_mode="antialiased")
    info["means2d"].retain_grad()
    ((r[0] * wr).sum() + (a[0, ..., 0] * wa).sum()).backward()
    gr = built.backward(fw, wr.numpy(), wa.numpy())
    errs = {}
    for name, want in (("means", p[0].grad), ("quats", p[1].grad), ("scales", p[2].grad), ("opacities", p[3].grad),
                       ("colors", p[4].grad), ("means2d", info["means2d"].grad[0]), ("absgrad", info["means2d"].absgrad[0])):
        assert_close(gr[name], want, rtol=1e-4, name=name)  # max_bad = 0
        errs[name] = rel_err(gr[name], want)
    record_cpu("c_vs_torch_oracle_backward", seed=seed, removed=removed, borderline_pixels=int(border.sum()), **errs)


def test_c_training_step_matches_torch_protocol(built):
    """ego_train_step == (torch oracle forward -> weight-map loss -> autograd -> absgrad -> torch Adam).  The FIRST step
    is compared strictly (cleaned scene, borderline pixels weightless: loss 1e-6, every element of the Adam update to the
    propagated tolerance); the following steps are a protocol check (alternating views, absgrad accumulation, step
    counts) -- after one Adam step two fp32 implementations no longer hold bit-identical parameters, so their
    borderline sets differ and the tolerance is the looser one stated below."""
    sc, _ = clean_scene(_scene(n=800, w=96, h=64, seed=5), [0, 1])
    n = sc.means.shape[0]
    lrs = {"means": 2e-3, "scales": 1e-4, "quats": 1e-3, "opacities": 0.03}
    tr = built.CpuTrainer(sc.means.numpy(), sc.log_scales.numpy(), sc.quats.numpy(), sc.logit_opacities.numpy(), lrs)
    P = {"means": torch.nn.Parameter(sc.means.clone()), "scales": torch.nn.Parameter(sc.log_scales.clone()),
         "quats": torch.nn.Parameter(sc.quats.clone()), "opacities": torch.nn.Parameter(sc.logit_opacities.clone())}
    opts = [torch.optim.Adam([P[k]], lr=lrs[k]) for k in P]
    absg = torch.zeros(n)
    init = {"means": sc.means, "scales": sc.log_scales, "quats": sc.quats, "opacities": sc.logit_opacities}
    for step, v in enumerate([0, 1, 0]):
        w = synth.weight_map("weighted", sc.gt[v])
        if step == 0:
            from tests.util import masked_weights, oracle_forward
            w = masked_weights(w, borderline_pixel_mask(oracle_forward(sc, v), sc.gt[v]))
        loss_c, M = tr.train_step(sc.viewmats[v].numpy(), sc.Ks[v].numpy(), sc.width, sc.height, sc.gt[v].numpy(), w.numpy())
        r, _, info = O.rasterization(P["means"], P["quats"], torch.exp(P["scales"]), torch.sigmoid(P["opacities"]).squeeze(-1),
                                     torch.ones(n, 3), sc.viewmats[v:v + 1], sc.Ks[v:v + 1], sc.width, sc.height,
                                     packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        loss = O.edge_step_loss(r[0, ..., 0], sc.gt[v], w)
        loss.backward()
        absg += info["means2d"].absgrad[0].norm(dim=-1)
        grads = {k: P[k].grad.detach().clone() for k in P}
        for o in opts:
            o.step()
            o.zero_grad()
        assert abs(loss_c - float(loss.detach())) <= (1e-5 if step == 0 else 2e-4) * abs(float(loss.detach())) and M == info["flatten_ids"].numel()
        if step == 0:  # the first Adam step is lr * g / (|g| + eps): every element to the propagated tolerance
            for name, mine in (("means", tr.means), ("scales", tr.log_scales), ("quats", tr.quats), ("opacities", tr.logit[:, None])):
                d_c, d_t = torch.from_numpy(mine) - init[name], P[name].data - init[name]
                g = grads[name].abs().double()
                tol = lrs[name] * (1e-4 + 1e-8 * 1e-4 * g.max() / (g + 1e-8) ** 2) + 1e-9
                assert bool(((d_c - d_t).abs().double() <= tol).all()), (name, float(((d_c - d_t).abs().double() - tol).max()))
            assert_close(tr.absgrads, absg, name="absgrads step 0")
    for name, mine in (("means", tr.means), ("scales", tr.log_scales), ("quats", tr.quats), ("opacities", tr.logit[:, None])):
        assert_close(torch.from_numpy(mine) - init[name], P[name].data - init[name], rtol=2e-3, max_bad=2e-2, name=f"delta {name}")
    assert_close(tr.absgrads, absg, max_bad=5e-3, name="absgrads")
