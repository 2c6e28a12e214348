"""World-size-2 test of the view-sharded data-parallel step on CPU (gloo backend).

The per-rank compute is the CPU oracle (the HIP path needs a GPU); what is under test is the
driver: view sharding, ONE all-reduce(sum) of the fused [N,12] buffer, identical Adam on every
rank.  Guarantee: all-reduced gradient == sum of the single-process per-view
gradients at the same parameters, and replicas stay bit-identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from edgegaussians_amd import dist as egdist
from edgegaussians_amd import synth
from oracle import ref_torch as O


class CpuOracleWorker:
    """GradWorker protocol on CPU tensors, same buffer layout as EdgeTrainer.grads."""

    def __init__(self, sc):
        self.sc = sc
        self.N = sc.means.shape[0]
        self.p = [sc.means.clone(), sc.quats.clone(), sc.log_scales.clone(), sc.logit_opacities.clone().view(-1)]
        self.m = [torch.zeros_like(t) for t in self.p]
        self.v = [torch.zeros_like(t) for t in self.p]
        self.lrs = [2e-3, 1e-3, 1e-4, 0.03]
        self.step = 0
        self.absgrads = torch.zeros(self.N)
        self.grads = torch.zeros(self.N * 12)
        self.announced = []

    def grad_views(self):
        N, g = self.N, self.grads
        return g[:3 * N].view(N, 3), g[3 * N:7 * N].view(N, 4), g[7 * N:10 * N].view(N, 3), g[10 * N:11 * N], g[11 * N:]

    def grad_step(self, view, wmap):
        sc, N = self.sc, self.N
        q = [t.clone().requires_grad_(True) for t in self.p]
        render, _, info = O.rasterization(
            means=q[0], quats=q[1], scales=torch.exp(q[2]), opacities=torch.sigmoid(q[3]), colors=torch.ones(N, 3),
            viewmats=sc.viewmats[view:view + 1], Ks=sc.Ks[view:view + 1], width=sc.width, height=sc.height,
            packed=False, absgrad=True, rasterize_mode="antialiased")
        info["means2d"].retain_grad()
        O.edge_step_loss(render[0, ..., 0], sc.gt[view], wmap).backward()
        for dst, src in zip(self.grad_views(), [t.grad for t in q] + [info["means2d"].absgrad[0].norm(dim=-1)]):
            dst.copy_(src)
        return self.grads

    def apply_adam(self, next_view=None):
        self.announced.append(next_view)  # (EdgeTrainer projects that view in the same launch; nothing to do here)
        self.step += 1
        gs = self.grad_views()
        for i in range(4):
            self.p[i], self.m[i], self.v[i] = O.adam_reference(self.p[i], gs[i], self.m[i], self.v[i], self.step,
                                                               self.lrs[i])
        self.absgrads += gs[4]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, _, w = egdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sc = synth.make_scene(300, 4, 64, 48, seed=2, spread_opacity=True, scale=0.03)
    wk = CpuOracleWorker(sc)
    dp = egdist.DataParallelStep(wk)
    wmaps = [synth.weight_map("weighted", sc.gt[v]) for v in range(4)]
    ok = True
    for step in range(2):
        view = egdist.view_for(step, rank, world, 4)
        # single-process ground truth: the sum over the step's view batch at the same parameters
        ref = CpuOracleWorker(sc)
        ref.p = [t.clone() for t in wk.p]
        want = torch.zeros_like(wk.grads)
        for rr in range(world):
            vv = egdist.view_for(step, rr, world, 4)
            want += ref.grad_step(vv, wmaps[vv]).clone()
        dp.step(view, wmaps[view], next_view=egdist.view_for(step + 1, rank, world, 4) if step == 0 else None)
        ok &= bool(torch.allclose(wk.grads, want, rtol=1e-5, atol=1e-9))
        # replicas identical: max |p - p_rank0| == 0 bit for bit
        for t in wk.p + [wk.absgrads]:
            ref_t = t.clone()
            dist.broadcast(ref_t, src=0)
            ok &= bool(torch.equal(ref_t, t))
    ok &= wk.announced == [egdist.view_for(1, rank, world, 4), None]
    # DataParallelStep.steps (round 4: K steps per call; natively eg_train_steps_dp on an EdgeTrainer with an RCCL
    # communicator): on a worker without the native leg it is exactly the loop of step() with the next view announced
    ok &= not dp.native_ready()
    views = [egdist.view_for(s_, rank, world, 4) for s_ in (2, 3)]
    n0 = len(wk.announced)
    dp.steps(views, [wmaps[v] for v in views], next_view=egdist.view_for(4, rank, world, 4))
    ok &= wk.announced[n0:] == [views[1], egdist.view_for(4, rank, world, 4)]
    for t in wk.p + [wk.absgrads]:
        ref_t = t.clone()
        dist.broadcast(ref_t, src=0)
        ok &= bool(torch.equal(ref_t, t))
    ret[rank] = ok
    dist.destroy_process_group()


def test_view_sharding_covers_every_view_once():
    for world in (1, 2, 4, 8):
        seen = [egdist.view_for(s, r, world, 50) for s in range(50 // world + 1) for r in range(world)]
        assert sorted(set(seen[:50])) == list(range(50)) and len(seen[:50]) == 50


@pytest.mark.timeout(600)
def test_allreduced_gradient_equals_sum_of_per_view_gradients():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ------------------------------------------------------------------ the same driver over EdgeTrainer on the GPU box
def _gpu_worker(rank, world, port, ret):
    """Two ranks sharing cuda:0 (gloo, gradients staged through the host: RCCL refuses two ranks on one device).
    Under test: DataParallelStep over the HIP path -- single views and C = 2 / 3 views per rank (batched launch
    sequences, the first half's all-reduce issued before the second half is computed)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from edgegaussians_amd import EdgeTrainer, LRSchedule
    torch.cuda.set_device(0)
    r, _, w = egdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    V = 8
    sc = synth.make_scene(3000, V, 160, 112, seed=4, spread_opacity=True, scale=0.02)
    sched = LRSchedule(scales_start=0, quats_start=0, opacities_start=0)
    mk = lambda: EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,  # noqa: E731
                             sc.width, sc.height, schedule=sched, seed=7)
    tr, ref = mk(), mk()
    tr.ensure_capacity()
    ref.ensure_capacity()
    dp = egdist.DataParallelStep(tr)
    wmaps = [synth.weight_map("weighted" if v % 2 else "whole", sc.gt[v]).cuda() for v in range(V)]
    ok, step = True, 0
    for Cn in (1, 1, 2, 3):
        # single-process ground truth at the replicas' current parameters: the sum over the step's whole view batch
        for name in ("means", "log_scales", "quats", "logit_opacities"):
            getattr(ref, name).copy_(getattr(tr, name))
        want = torch.zeros_like(tr.grads)
        for rr in range(world):
            for slot in range(Cn):
                vv = egdist.view_for(step, rr, world, V, Cn, slot)
                want += ref.grad_step(vv, wmaps[vv])
        mine = [egdist.view_for(step, rank, world, V, Cn, slot) for slot in range(Cn)]
        if Cn == 1:
            # the rank announces its next view (Adam then projects it in the same launch); after step 1 the announced
            # view is NOT what follows (a batched step does): the pre-projection must be dropped cleanly
            dp.step(mine[0], wmaps[mine[0]], next_view=egdist.view_for(step + 1, rank, world, V))
        else:
            dp.step(mine, [wmaps[v] for v in mine])
        got = tr.grads.clone()
        scale = float(want.abs().max())
        ok &= bool(((got - want).abs() <= 1e-5 * scale + 1e-5 * want.abs()).all())
        for t in (tr.means, tr.log_scales, tr.quats, tr.logit_opacities, tr.absgrads, tr.adam_m):
            h = t.detach().cpu()
            h0 = h.clone()
            dist.broadcast(h0, src=0)
            ok &= bool(torch.equal(h0, h))  # replicas bit-identical
        step += 1
    ok &= tr.adam_step == 4 and not tr.overflowed()
    # a densify event draws its noise from the seeded per-trainer generator: replicas must stay identical
    tr.absgrads.copy_(torch.arange(tr.N, device="cuda").float())
    tr.duplicate_high_pos_gradients(0.9, 3, 0.05)
    h = tr.means.detach().cpu()
    h0 = h.clone()
    dist.broadcast(h0, src=0)
    ok &= bool(torch.equal(h0, h)) and tr.N > 3000
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_data_parallel_driver_over_the_hip_path_two_ranks_one_gpu():
    assert torch.cuda.is_available()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ------------------------------------------------------------------ train() under data parallelism (two ranks, one GPU)
def _dp_train_cfg():
    optim = {"means": {"start_lr": 2e-3, "milestones": [], "gamma": 1.0}, "scales": {"start_lr": 1e-4, "start_at_epoch": 0},
             "quats": {"start_lr": 1e-3, "start_at_epoch": 0}, "opacities": {"start_lr": 0.5, "start_at_epoch": 0}}
    training_cfg = {"num_epochs": 4, "optim": optim, "loss": {
        "orientation_losses": {"start_dir_loss_at_epoch": 1, "start_ratio_loss_at_epoch": 1, "dir_loss_num_nn": 5,
                               "dir_loss_scale_factor": 0.01, "ratio_loss_scale_factor": 0.01},
        "projection_losses": {"lambda_annealing": "constant", "lambda_start": 1, "lambda_end": 1,
                              "loss_before_alternating": "whole", "less_freq_loss": "bg_edge_ratio",
                              "more_freq_loss": "whole", "start_alternating_at_epoch": 0,
                              "bg_edge_pixel_ratio_annealing": "constant", "bg_edge_pixel_ratio_start": 1,
                              "bg_edge_pixel_ratio_end": 1, "sampling_whole_num_epochs_ratio": 3}}}
    model_cfg = {"if_duplicate_high_pos_grad": True, "dup_high_pos_grads_at_epoch": [2], "dup_threshold_type": "absolute",
                 "dup_threshold_value": 0.3, "dup_factor": 2, "init_dup_rand_noise_scale": 0.01,
                 "if_cull_low_opacity": False, "if_cull_gaussians_not_projecting": False}
    return model_cfg, training_cfg


def _dp_train_worker(rank, world, port, ret):
    """`train()` under DataParallelStep, two ranks sharing cuda:0 (gloo): across a duplication event, regulariser steps
    and a capacity crossing (buffers sized without slack while the opacities run up: some rank's step overflows, the
    collective read-back makes BOTH ranks grow and replay).  Replicas bit-identical; history equal to the
    single-process run that takes the same two views per optimizer step as one batched launch sequence."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from edgegaussians_amd import EdgeTrainer, train
    torch.cuda.set_device(0)
    r, _, w = egdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    V = 6
    sc = synth.make_scene(6000, V, 200, 136, seed=3, spread_opacity=False, scale=0.02, anisotropy=5.0)  # (the overflow scene of test_gpu_parity)
    model_cfg, training_cfg = _dp_train_cfg()

    def mk():
        t = EdgeTrainer(sc.means, sc.log_scales, sc.quats, sc.logit_opacities, sc.viewmats, sc.Ks, sc.gt,
                        sc.width, sc.height, seed=7)
        t.ensure_capacity(slack=1.0)
        t._alloc_isect(t.m_max_seen + 64, (t.max_tile_seen // 128 + 1) * 128)  # no slack: the opacity ramp overflows it
        t.gt.fill_(1.0)  # a target that wants everything opaque
        return t

    order = lambda e: [(e + k) % V for k in range(V)]  # noqa: E731
    tr = mk()
    sized = (tr.capacity, tr.seg_cap, tr.max_tile_seen, tr.m_max_seen)
    dp = egdist.DataParallelStep(tr)
    n_seen = []
    # (read-backs only before the duplication of epoch 2 and at the end: nine optimizer steps at opacity lr 0.5 lie
    # between the exact sizing and the first look at the overflow flag)
    hist = train(tr, model_cfg, training_cfg, order, views_per_step=2, dp=dp, on_epoch=lambda e, l, n: n_seen.append(n),
                 sync_every=10)
    checks = {"hist": len(hist) == 4 and all(map(lambda x: x == x and abs(x) < 1e9, hist)),
              "crossing": tr.overflow_events >= 1 and not tr.overflowed(),   # the capacity crossing happened and was repaired
              "dup": tr.N > 6000 and n_seen[-1] == tr.N}                      # the duplication event happened
    for name in ("means", "log_scales", "quats", "logit_opacities", "absgrads", "adam_m", "adam_v"):
        h = getattr(tr, name).detach().cpu()
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([h.shape[0]], dtype=torch.int64))
        same = all(int(x) == h.shape[0] for x in sizes)
        if same:
            h0 = h.clone()
            dist.broadcast(h0, src=0)
            same = bool(torch.equal(h0, h))                                   # replicas bit-identical
            if not same:
                checks[f"diff_{name}"] = (int((h0 != h).sum()), float((h0 - h).abs().max()))
        checks[f"replica_{name}"] = same
    hh = torch.tensor(hist, dtype=torch.float64)
    h0 = hh.clone()
    dist.broadcast(h0, src=0)
    checks["same_history"] = bool(torch.equal(h0, hh))                        # every rank reports the same history
    ok = all(v for k, v in checks.items() if not k.startswith("diff_"))
    ret[f"checks{rank}"] = dict(checks)
    if rank == 0:
        ref = mk()
        hist_ref = train(ref, model_cfg, training_cfg, order, views_per_step=2, sync_every=10)
        ok &= ref.N == tr.N
        ok &= all(abs(a - b) <= 2e-3 * abs(b) for a, b in zip(hist, hist_ref))
        ret["hist"] = (list(hist), list(hist_ref), tr.overflow_events, ref.overflow_events, tr.rewalk_misses, sized,
                       (tr.capacity, tr.seg_cap, tr.max_tile_seen, tr.last_m()),
                       float(torch.sigmoid(tr.logit_opacities).mean()))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_data_parallel_training_survives_events_and_a_capacity_crossing():
    assert torch.cuda.is_available()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_train_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret.get(0) is True and ret.get(1) is True, dict(ret)


def test_reduce_words_over_gloo():
    """The two small collectives of a data-parallel read-back: flags by max, loss sums by sum."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reduce_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _reduce_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    egdist.init_from_env("gloo")

    class W:  # (no journal: a bare GradWorker)
        pass
    dp = egdist.DataParallelStep(W())
    ints, floats = dp.reduce_words([rank, 1 - rank, 7], [0.5 + rank, 2.0])
    ret[rank] = ints == [1, 1, 7] and floats == [2.0, 4.0] and dp.reduce_words([], []) == ([], [])
    dist.destroy_process_group()
