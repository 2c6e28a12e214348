"""The two CPU oracles against each other: oracle/eg_oracle.c (plain C, per-pixel sequential walk,
hand-derived backward with transmittance recovery) vs oracle/ref_torch.py (dense tensors, autograd).
They share no code, so agreement pins the restated semantics from two independent derivations."""
import numpy as np
import pytest
import torch

from edgegaussians_amd import synth
from oracle import c_oracle as CO
from oracle import ref_torch as O
from tests.util import assert_close, rel_err


@pytest.fixture(scope="module")
def built():
    CO.build()
    return CO


def _scene(n=1200, w=112, h=80, seed=0, scale=0.02):
    return synth.make_scene(n, 2, w, h, seed=seed, spread_opacity=True, scale=scale)


@pytest.mark.parametrize("mode", ["antialiased", "classic"])
def test_forward_matches_torch_oracle(built, mode):
    sc = _scene()
    scales, opac = torch.exp(sc.log_scales), torch.sigmoid(sc.logit_opacities).squeeze(-1)
    colors = torch.rand(sc.means.shape[0], 3, generator=torch.Generator().manual_seed(1))
    r, a, info = O.rasterization(sc.means, sc.quats, scales, opac, colors, sc.viewmats[:1], sc.Ks[:1], sc.width,
                                 sc.height, packed=False, rasterize_mode=mode)
    fw = built.rasterize(sc.means.numpy(), sc.quats.numpy(), scales.numpy(), opac.numpy(), colors.numpy(),
                         sc.viewmats[0].numpy(), sc.Ks[0].numpy(), sc.width, sc.height, antialiased=(mode == "antialiased"))
    ro = info["radii"][0].numpy()
    assert (fw["radii"] != ro).mean() < 2e-3
    same = fw["radii"] == ro
    assert_close(fw["means2d"][same], info["means2d"][0].detach().numpy()[same], name="means2d")
    assert_close(fw["conics"][same], info["conics"][0].detach().numpy()[same], name="conics")
    assert_close(fw["comps"][same], (info["opacities"][0] / opac.clamp_min(1e-12)).numpy()[same] if mode == "antialiased"
                 else fw["comps"][same], name="comps")
    assert_close(fw["render"], r[0], max_bad=2e-3, name="render")
    assert_close(fw["alphas"], a[0, ..., 0], max_bad=2e-3, name="alpha")
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


def test_backward_general_colours_matches_autograd(built):
    sc = _scene(n=900, w=96, h=64, seed=3)
    N = sc.means.shape[0]
    g = torch.Generator().manual_seed(2)
    colors0 = torch.rand(N, 3, generator=g)
    wr, wa = torch.rand(sc.height, sc.width, 3, generator=g), torch.rand(sc.height, sc.width, generator=g) * 0.1
    p = [t.clone().requires_grad_(True) for t in (sc.means, sc.quats, torch.exp(sc.log_scales),
                                                  torch.sigmoid(sc.logit_opacities).squeeze(-1), colors0)]
    r, a, info = O.rasterization(*p, sc.viewmats[:1], sc.Ks[:1], sc.width, sc.height, packed=False, absgrad=True,
                                 rasterize_mode="antialiased")
    info["means2d"].retain_grad()
    ((r[0] * wr).sum() + (a[0, ..., 0] * wa).sum()).backward()
    fw = built.rasterize(*[t.detach().numpy() for t in p], sc.viewmats[0].numpy(), sc.Ks[0].numpy(), sc.width, sc.height)
    gr = built.backward(fw, wr.numpy(), wa.numpy())
    for name, want in (("means", p[0].grad), ("quats", p[1].grad), ("scales", p[2].grad), ("opacities", p[3].grad),
                       ("colors", p[4].grad), ("means2d", info["means2d"].grad[0]), ("absgrad", info["means2d"].absgrad[0])):
        assert rel_err(gr[name], want) < 2e-3, (name, rel_err(gr[name], want))
        assert_close(gr[name], want, rtol=2e-4, max_bad=5e-3, name=name)


def test_c_training_step_matches_torch_protocol(built):
    """ego_train_step == (torch oracle forward -> weight-map loss -> autograd -> absgrad -> torch Adam)."""
    sc = _scene(n=800, w=96, h=64, seed=5)
    lrs = {"means": 2e-3, "scales": 1e-4, "quats": 1e-3, "opacities": 0.03}
    tr = built.CpuTrainer(sc.means.numpy(), sc.log_scales.numpy(), sc.quats.numpy(), sc.logit_opacities.numpy(), lrs)
    P = {"means": torch.nn.Parameter(sc.means.clone()), "scales": torch.nn.Parameter(sc.log_scales.clone()),
         "quats": torch.nn.Parameter(sc.quats.clone()), "opacities": torch.nn.Parameter(sc.logit_opacities.clone())}
    opts = [torch.optim.Adam([P[k]], lr=lrs[k]) for k in P]
    absg = torch.zeros(800)
    for step, v in enumerate([0, 1, 0]):
        w = synth.weight_map("weighted", sc.gt[v])
        loss_c, M = tr.train_step(sc.viewmats[v].numpy(), sc.Ks[v].numpy(), sc.width, sc.height, sc.gt[v].numpy(), w.numpy())
        r, _, info = O.rasterization(P["means"], P["quats"], torch.exp(P["scales"]), torch.sigmoid(P["opacities"]).squeeze(-1),
                                     torch.ones(800, 3), sc.viewmats[v:v + 1], sc.Ks[v:v + 1], sc.width, sc.height,
                                     packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        loss = O.edge_step_loss(r[0, ..., 0], sc.gt[v], w)
        loss.backward()
        absg += info["means2d"].absgrad[0].norm(dim=-1)
        for o in opts:
            o.step()
            o.zero_grad()
        assert abs(loss_c - float(loss)) <= 2e-4 * abs(float(loss)) and M == info["flatten_ids"].numel()
    for name, mine in (("means", tr.means), ("scales", tr.log_scales), ("quats", tr.quats), ("opacities", tr.logit[:, None])):
        init = {"means": sc.means, "scales": sc.log_scales, "quats": sc.quats, "opacities": sc.logit_opacities}[name]
        assert_close(torch.from_numpy(mine) - init, P[name].data - init, rtol=2e-3, max_bad=2e-2, name=f"delta {name}")
    assert_close(tr.absgrads, absg, max_bad=5e-3, name="absgrads")
