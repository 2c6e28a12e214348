"""The reference's epoch / iteration protocol, driven on the fused device step.

Host-side mirror of `train()` and `train_epoch()` (`/root/reference/train_gaussians.py:17-222`): same
config dictionaries (the `model` and `training` sections of `configs/*.json`), same calendar, same
quirks -- but every iteration is one `EdgeTrainer.train_step` enqueue and the only host syncs are the
ones the semantics require (the regulariser lambda needs the loss values, train_gaussians.py:113,125;
densify/cull events change N).

Line map:
    bg_edge_pixel_ratio / lambda_projection annealing   train_utils.py:28-45, train_gaussians.py:29-34
    strategy alternation (model.step % ratio)           train_gaussians.py:59-77
    per-view iteration                                  :81-106      -> EdgeTrainer.train_step
    direction / ratio regulariser every 5th step        :108-131     -> EdgeTrainer.regulariser_step
    schedulers step once per epoch                      :183-184     -> LRSchedule.at(epoch)
    duplicate / cull calendar, absgrad reset            :186-219     -> EdgeTrainer.duplicate*/cull*
The view order is the caller's (`view_order(epoch) -> iterable of view indices`): the reference uses
`DataLoader(shuffle=True)` on the default CPU generator (train_gaussians.py:311).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch

from .trainer import EdgeTrainer, LRSchedule


def _anneal(cfg: Dict, key: str, step: int, max_steps: int) -> float:
    kind = cfg[f"{key}_annealing"]
    if kind == "constant":
        return cfg[f"{key}_start"]
    if kind == "linear":
        return cfg[f"{key}_start"] + (cfg[f"{key}_end"] - cfg[f"{key}_start"]) * step / max_steps
    raise ValueError(f"Unsupported {key}_annealing: {kind}")


def train_epoch(tr: EdgeTrainer, views: Iterable[int], epoch: int, num_epochs: int, projection_cfg: Dict,
                orientation_cfg: Dict, edge_threshold: float = 0.5,
                generator: Optional[torch.Generator] = None, read_back: bool = True,
                views_per_step: int = 1, dp=None) -> Optional[float]:
    """train_gaussians.py:17-141 for one epoch; returns the average projection loss -- or, with read_back=False,
    parks the epoch's loss sum on the device (EdgeTrainer.mark_epoch, no host sync) and returns the number of
    iterations; the caller collects the sums of several epochs with one `pop_losses()`.

    views_per_step = C > 1 (a THROUGHPUT mode, SURVEY 8e/8f: the reference steps after every view): the epoch's views
    are taken C at a time and every batch is ONE optimizer step on the sum of its views' gradients (a last, short batch
    is filled up from the front of the epoch's order) -- on one GPU as a batched launch sequence
    (`EdgeTrainer.train_step_batched`), or with `dp` (a `dist.DataParallelStep` over `tr`, world size P dividing C)
    sharded over the ranks: rank r rasterises views r C/P .. (r + 1) C/P - 1 of every batch, one all-reduce of the fused
    [N,12] gradient buffer, the identical Adam on every rank.  Both forms follow the same trajectory up to the rounding
    of the gradient sum; strategies, weight-map draws and the regulariser cadence count VIEWS exactly as with C = 1."""
    C = int(views_per_step)
    world = dp.world if dp is not None else 1
    rank = dp.rank if dp is not None else 0
    if C < 1 or C % world:
        raise ValueError(f"views_per_step={C} must be a positive multiple of the world size {world}")
    if C > 1 or dp is not None:
        return _train_epoch_batched(tr, list(views), epoch, num_epochs, projection_cfg, orientation_cfg, edge_threshold,
                                    generator, read_back, C, dp, world, rank)
    ratio = _anneal(projection_cfg, "bg_edge_pixel_ratio", epoch, num_epochs)
    tr.loss_scale = float(_anneal({"lambda_annealing": projection_cfg["lambda_annealing"],
                                   "lambda_start": projection_cfg["lambda_start"],
                                   "lambda_end": projection_cfg["lambda_end"]}, "lambda", epoch, num_epochs))
    tr.epoch = epoch
    alternate = epoch > projection_cfg["start_alternating_at_epoch"]
    strategy = projection_cfg["loss_before_alternating"]
    apply_dir = epoch > orientation_cfg["start_dir_loss_at_epoch"]
    apply_ratio = epoch > orientation_cfg["start_ratio_loss_at_epoch"]
    period = projection_cfg["sampling_whole_num_epochs_ratio"]
    # The iterations between two regulariser steps go to the device as ONE native enqueue (EdgeTrainer.train_steps);
    # the regulariser weight lambda = (running loss sum of this epoch) * factor / loss is formed on the device from
    # the loss accumulator, so the epoch runs without a single host sync; the one read-back at its end also checks
    # the sticky overflow flag (and replays the epoch from its journal if some step dropped intersections).
    n, step_no = 0, tr.step
    pend_v, pend_w = [], []

    def flush_steps():
        if pend_v:
            if generator is not None:
                maps = list(pend_w)
            elif hasattr(tr, "weight_maps"):  # (strategies were parked: the run's maps by ONE native call -- same draws, same order)
                maps = tr.weight_maps(list(pend_v), list(pend_w), ratio, edge_threshold)
            else:
                maps = [tr.weight_map(v, st, ratio, None, edge_threshold) for v, st in zip(pend_v, pend_w)]
            tr.train_steps(list(pend_v), maps)
            pend_v.clear()
            pend_w.clear()

    for idx in views:
        if alternate:
            strategy = projection_cfg["less_freq_loss"] if step_no % period == 0 else projection_cfg["more_freq_loss"]
        pend_v.append(int(idx))
        # (device-side draws: park the strategy, flush_steps draws the run's maps together; a host generator -- the
        # reference's CPU randperm -- is advanced by drawing, so it draws here)
        pend_w.append(strategy if generator is None else tr.weight_map(idx, strategy, ratio, generator, edge_threshold))
        n += 1
        step_no += 1
        if (apply_dir or apply_ratio) and step_no % 5 == 0:
            flush_steps()
            # (the device accumulator holds the UNSCALED losses of the epoch so far, train_gaussians.py:99,113,125)
            if apply_dir:
                tr.regulariser_step("direction", None, orientation_cfg["dir_loss_scale_factor"],
                                    orientation_cfg["dir_loss_num_nn"],
                                    orientation_cfg.get("dir_loss_enforce_method", "enforce_full"), want_value=False)
            if apply_ratio:
                tr.regulariser_step("ratio", None, orientation_cfg["ratio_loss_scale_factor"], want_value=False)
    flush_steps()
    if not read_back:
        tr.mark_epoch()
        return n
    return tr.pop_loss() / max(n, 1)


def _train_epoch_batched(tr, views, epoch, num_epochs, projection_cfg, orientation_cfg, edge_threshold, generator,
                         read_back, C, dp, world, rank):
    ratio = _anneal(projection_cfg, "bg_edge_pixel_ratio", epoch, num_epochs)
    tr.loss_scale = float(_anneal({"lambda_annealing": projection_cfg["lambda_annealing"],
                                   "lambda_start": projection_cfg["lambda_start"],
                                   "lambda_end": projection_cfg["lambda_end"]}, "lambda", epoch, num_epochs))
    tr.epoch = epoch
    alternate = epoch > projection_cfg["start_alternating_at_epoch"]
    strategy = projection_cfg["loss_before_alternating"]
    apply_dir = epoch > orientation_cfg["start_dir_loss_at_epoch"]
    apply_ratio = epoch > orientation_cfg["start_ratio_loss_at_epoch"]
    period = projection_cfg["sampling_whole_num_epochs_ratio"]
    if len(views) % C:
        views = views + views[:C - len(views) % C]
    per_rank = C // world
    n, step_no = 0, tr.step if dp is None else getattr(tr, "_dp_views_seen", tr.step)
    native = dp is not None and per_rank == 1 and getattr(dp, "native_ready", lambda: False)()
    pend_v, pend_w = [], []
    for b0 in range(0, len(views), C):
        batch = views[b0:b0 + C]
        mine_v, mine_w = [], []
        for k, idx in enumerate(batch):
            if alternate:
                strategy = projection_cfg["less_freq_loss"] if (step_no + k) % period == 0 else projection_cfg["more_freq_loss"]
            if k // per_rank == rank:
                mine_v.append(int(idx))
                mine_w.append(tr.weight_map(idx, strategy, ratio, generator, edge_threshold))
            elif strategy == "bg_edge_ratio" and generator is None:
                tr.skip_weight_map_draw()  # (another rank's view: keep the draw sequence in step)
            elif strategy == "bg_edge_ratio":
                tr.weight_map(idx, strategy, ratio, generator, edge_threshold)  # (a host generator must be advanced by drawing)
        if dp is None:
            tr.train_step_batched(mine_v, mine_w)
        elif per_rank == 1 and native:
            pend_v.append(mine_v[0])  # (the steps between two regulariser steps go out as ONE native enqueue:
            pend_w.append(mine_w[0])  #  DataParallelStep.steps -> eg_train_steps_dp)
        elif per_rank == 1:
            dp.step(mine_v[0], mine_w[0])
        else:
            dp.step(mine_v, mine_w)
        n += C
        crossed = (step_no + C) // 5 > step_no // 5  # (a multiple of five views was passed: train_gaussians.py:108)
        step_no += C
        if pend_v and (((apply_dir or apply_ratio) and crossed) or b0 + C >= len(views)):
            dp.steps(pend_v, pend_w)
            pend_v, pend_w = [], []
        if (apply_dir or apply_ratio) and crossed:
            if apply_dir:
                tr.regulariser_step("direction", None, orientation_cfg["dir_loss_scale_factor"],
                                    orientation_cfg["dir_loss_num_nn"],
                                    orientation_cfg.get("dir_loss_enforce_method", "enforce_full"), want_value=False)
            if apply_ratio:
                tr.regulariser_step("ratio", None, orientation_cfg["ratio_loss_scale_factor"], want_value=False)
    if dp is not None:
        tr._dp_views_seen = step_no  # (tr.step counts this rank's views only)
    if not read_back:
        tr.mark_epoch()
        return n
    return tr.pop_loss() / max(n, 1)


def train(tr: EdgeTrainer, model_cfg: Dict, training_cfg: Dict, view_order: Callable[[int], Iterable[int]],
          edge_masks_u8: Optional[torch.Tensor] = None, on_epoch: Optional[Callable[[int, float, int], None]] = None,
          generator: Optional[torch.Generator] = None, num_epochs: Optional[int] = None,
          sync_every: Optional[int] = None, views_per_step: int = 1, dp=None) -> List[float]:
    """train_gaussians.py:144-222.  `model_cfg` / `training_cfg` are the reference's JSON sections;
    `edge_masks_u8` [V,H,W] (gt >= threshold) is needed only for the not-projecting cull.

    The host reads the device back (loss sums, sticky overflow flag, launch-shape hints: ONE small copy) only every
    `sync_every` epochs, before every densify / cull event and at the end; in between the epochs are enqueued
    back to back with their loss sums parked on the device, so the host prepares epoch e + 1 while the device still
    runs epoch e.  `on_epoch(epoch, avg_loss, N)` is therefore called late -- in order, at the next read-back.
    sync_every = 1 is the reference's cadence (one read-back per epoch) and the default when `on_epoch` is given (a
    callback that snapshots the trainer then sees the state at the end of ITS epoch); without a callback the default is
    8.  A callback that only logs may pass a larger `sync_every` explicitly.

    views_per_step / dp: see `train_epoch` -- C views per optimizer step, on one GPU or sharded over the ranks of a
    `dist.DataParallelStep`.  Under `dp` every rank runs this same loop: the read-backs are collective (sticky overflow /
    missed-stop flags by max, so that all ranks replay the same steps; loss sums by sum, so that every rank returns the
    same history), densify / cull decisions come from the all-reduced absgrads and the seeded noise, and the replicas
    stay bit-identical."""
    if sync_every is None:
        sync_every = 1 if on_epoch is not None else 8
    loss_cfg = training_cfg["loss"]
    proj_cfg, orient_cfg = loss_cfg["projection_losses"], loss_cfg["orientation_losses"]
    num_epochs = training_cfg["num_epochs"] if num_epochs is None else num_epochs
    tr.schedule = LRSchedule.from_config(training_cfg["optim"])
    get = model_cfg.get
    thr = get("edge_detection_threshold", 0.5)
    if tr.capacity == 0:
        tr.ensure_capacity()
    history: List[float] = []
    parked: List = []  # (epoch, iterations) of the epochs whose loss sums are still on the device
    ready: List = []  # [epoch, N] read back, on_epoch not yet called (it reports N AFTER the epoch's events)
    events = set()
    for flag, key, default in (("if_duplicate_high_pos_grad", "dup_high_pos_grads_at_epoch", True),
                               ("if_cull_gaussians_not_projecting", "cull_gaussians_not_projecting_at_epoch", True),
                               ("if_cull_low_opacity", "cull_opacity_at_epoch", True),
                               ("if_cull_wayward", "cull_wayward_at_epoch", False)):  # (dataclass defaults, edge_gs.py:20-54)
        if get(flag, default):
            events.update(get(key, []))
    if get("if_reset_opacity", False):
        # train_gaussians.py:210-211 calls model.reset_opacities(optimizers), which takes no argument (edge_gs.py:425):
        # the reference dies with a TypeError at the first such epoch; every shipped config spells the key
        # "if reset_opacity" (configs/*.json:37), which the dataclass never sees
        raise NotImplementedError("if_reset_opacity=True: the reference raises TypeError at reset_opacity_at_epoch "
                                  "(train_gaussians.py:211 passes an argument edge_gs.py:425 does not take)")
    reset_value = get("reset_opacity_value", 0.08)  # the clamp of every cull (edge_gs.py:416-417,425-429)

    def read_back():
        sums, _rest = tr.pop_losses()
        assert len(sums) == len(parked)
        for (e, n), sm in zip(parked, sums):
            history.append(sm / max(n, 1))
            ready.append([e, tr.N])
        ready[-1][1] = None  # the epoch that just ran: its events (below) may still change N
        parked.clear()

    for epoch in range(num_epochs):
        n = train_epoch(tr, view_order(epoch), epoch, num_epochs, proj_cfg, orient_cfg, thr, generator, read_back=False,
                        views_per_step=views_per_step, dp=dp)
        parked.append((epoch, n))
        # (the trainer parks at most 64 epoch sums on the device between two read-backs)
        # (the journal's size is a RANK-LOCAL quantity -- a rank journals the weight maps of its own views only, and with
        # period-2 draws on two ranks every fresh `bg_edge_ratio` map lands on the same one -- while a read-back is a
        # collective: under data parallelism the ranks agree on the decision first, or one of them would enter the
        # read-back's all-reduce while the others go on to the next epoch's gradient all-reduce)
        big = len(tr._journal) > 4096 or tr.journal_bytes() > (1 << 29)
        if dp is not None and getattr(dp, "world", 1) > 1:
            big = bool(dp.reduce_words([int(big)], [])[0][0])
        if len(parked) >= max(1, min(sync_every, 60)) or epoch in events or epoch == num_epochs - 1 or big:
            read_back()
        changed = False
        if get("if_duplicate_high_pos_grad", True) and epoch in get("dup_high_pos_grads_at_epoch", []):
            kind = get("dup_threshold_type", "percentile")
            if kind not in ("absolute", "percentile_top"):
                # the reference has exactly these two branches (edge_gs.py:559-573); any other value -- the
                # dataclass default "percentile" included -- dies there with an unbound `dup_mask`
                raise NotImplementedError(f"dup_threshold_type={kind!r}: the reference defines only 'absolute' "
                                          "and 'percentile_top' (edge_gs.py:559-573)")
            tr.duplicate_high_pos_gradients(get("dup_threshold_value", 0.95), get("dup_factor", 2),
                                            get("init_dup_rand_noise_scale", 0.05), threshold_type=kind)
            changed = True
        if get("if_cull_gaussians_not_projecting", True) and epoch in get("cull_gaussians_not_projecting_at_epoch", []):
            if edge_masks_u8 is None:
                edge_masks_u8 = (tr.gt >= thr).to(torch.uint8)
            tr.cull_not_projecting(edge_masks_u8, get("cull_gaussians_not_projecting_threshold", 0.35),
                                   reset_opacity_value=reset_value)
            changed = True
        if get("if_cull_low_opacity", True) and epoch in get("cull_opacity_at_epoch", []):
            if get("cull_opacity_type", "absolute") == "absolute":
                tr.cull_opacity(get("cull_opacity_value", 0.05), reset_opacity_value=reset_value)
            else:
                q = torch.quantile(torch.sigmoid(tr.logit_opacities), get("cull_opacity_value", 0.05))
                tr.cull(torch.sigmoid(tr.logit_opacities) < q, reset_value)
            changed = True
        # cull_wayward computes a mask and never applies it (edge_gs.py:498-542) -- but its epochs still set
        # reset_absgrads (train_gaussians.py:204-208,218-219; configs/Replica.json:15,20 turns it on at epochs 45
        # and 300): the accumulator that decides the next duplication restarts there.  (update_nn is moot here: the
        # neighbour table is rebuilt by every direction-regulariser step.)
        wayward = get("if_cull_wayward", False) and epoch in get("cull_wayward_at_epoch", [])
        if changed or wayward:
            if tr._journal:
                tr.flush()
            tr.reset_absgrads()          # train_gaussians.py:218-219
        if changed:
            if tr.spatial_order:
                tr.spatial_sort()        # new rows were appended / rows were dropped: restore the layout
            tr.ensure_capacity()         # N changed: re-size the isect buffers (one count-only sweep)
        if on_epoch:
            for e, n_e in ready:
                on_epoch(e, history[e], tr.N if n_e is None else n_e)
        ready.clear()
    return history
