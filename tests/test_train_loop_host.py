"""Host logic of the epoch driver (edgegaussians_amd/train_loop.py, the mirror of train_gaussians.py:17-222) on a
recording stand-in for EdgeTrainer: the calendar, the strategy alternation, the regulariser cadence, the batching of
iterations into native runs, and the read-back cadence (loss sums parked on the device, one read-back per
`sync_every` epochs, before every densify / cull event and at the end) with `on_epoch` reporting each epoch's average
loss and the Gaussian count AFTER that epoch's events.  No GPU, no native library."""
import json
import os

import pytest

from edgegaussians_amd import train_loop
from edgegaussians_amd.trainer import LRSchedule


class FakeTrainer:
    """Loss of a step = 1 + 0.001 * (global step index); densify doubles N, culls remove 10 %."""

    def __init__(self, n=100):
        self.N, self.capacity, self.spatial_order = n, 1, True
        self.step, self.epoch, self.loss_scale = 0, 0, 1.0
        self.schedule = None
        self._journal = []
        self.acc, self.marks = 0.0, []
        self.log = []          # ("steps", epoch, views, strategies) | ("reg", kind) | ("sync",) | ("dup",) ...
        self.syncs = 0

    # -- what train_epoch uses
    def ensure_capacity(self):
        self.log.append(("capacity", self.N))

    def weight_map(self, view, strategy, ratio, generator, threshold):
        return (strategy, round(ratio, 6))

    def train_steps(self, views, wmaps):
        for _ in views:
            self.acc += 1.0 + 0.001 * self.step
            self.step += 1
        self._journal.extend(views)
        self.log.append(("steps", self.epoch, list(views), [w[0] for w in wmaps], self.loss_scale))

    def train_step_batched(self, views, wmaps):
        self.acc += sum(1.0 + 0.001 * (self.step + i) for i in range(len(views)))
        self.step += len(views)
        self._journal.append(tuple(views))
        self.log.append(("batch", self.epoch, list(views), [w[0] for w in wmaps], self.loss_scale))

    def skip_weight_map_draw(self):
        self.log.append(("skip_draw", self.epoch))

    def regulariser_step(self, kind, avg_loss_sum, scale_factor, *a, **k):
        assert avg_loss_sum is None  # lambda comes from the device accumulator
        self.log.append(("reg", kind, self.step, scale_factor))

    def mark_epoch(self):
        self.marks.append(self.acc)
        self.acc = 0.0
        return len(self.marks) - 1

    def pop_losses(self):
        self.syncs += 1
        self.log.append(("sync", self.epoch))
        m, rest = self.marks, self.acc
        self.marks, self.acc, self._journal = [], 0.0, []
        return m, rest

    def pop_loss(self):
        m, rest = self.pop_losses()
        return sum(m) + rest

    # -- events
    def duplicate_high_pos_gradients(self, *a, **k):
        assert not self.marks, "events run on a read-back state"
        self.N *= 2
        self.log.append(("dup", self.epoch))

    def cull_opacity(self, v, reset_opacity_value=0.08):
        assert not self.marks
        self.N -= self.N // 10
        self.log.append(("cull_opacity", self.epoch, v, reset_opacity_value))

    def cull_not_projecting(self, masks, thr, reset_opacity_value=0.08):
        assert not self.marks
        self.N -= self.N // 10
        self.log.append(("cull_np", self.epoch, thr, reset_opacity_value))

    def flush(self):
        self._journal = []

    def journal_bytes(self):
        return 0

    def reset_absgrads(self):
        self.log.append(("reset_absgrads", self.epoch))

    def spatial_sort(self):
        self.log.append(("sort", self.epoch))


@pytest.fixture
def cfg(golden_dir):
    return json.load(open(os.path.join(golden_dir, "abc_train_config.json")))  # configs/ABC_DexiNed.json, parsed


def _run(cfg, sync_every, epochs=60, views=7):
    model_cfg, training_cfg = dict(cfg["model"]), json.loads(json.dumps(cfg["training"]))
    model_cfg.update(dup_high_pos_grads_at_epoch=[10, 30], cull_opacity_at_epoch=[20],
                     cull_gaussians_not_projecting_at_epoch=[30, 45])
    ol, pl = training_cfg["loss"]["orientation_losses"], training_cfg["loss"]["projection_losses"]
    ol["start_dir_loss_at_epoch"], ol["start_ratio_loss_at_epoch"] = 40, 40
    pl["start_alternating_at_epoch"] = 5
    tr = FakeTrainer()
    seen = []
    hist = train_loop.train(tr, model_cfg, training_cfg, lambda e: [(e + i) % views for i in range(views)],
                            edge_masks_u8=object(), on_epoch=lambda e, l, n: seen.append((e, l, n)),
                            num_epochs=epochs, sync_every=sync_every)
    return tr, hist, seen


@pytest.mark.parametrize("sync_every", [1, 4, 8, 1000])
def test_history_and_on_epoch_do_not_depend_on_the_read_back_cadence(cfg, sync_every):
    tr, hist, seen = _run(cfg, sync_every)
    views, epochs = 7, 60
    # average loss of epoch e: steps e*7 .. e*7+6 of the fake's loss law
    for e in range(epochs):
        want = sum(1.0 + 0.001 * (e * views + i) for i in range(views)) / views
        assert abs(hist[e] - want) < 1e-12
    assert [s[0] for s in seen] == list(range(epochs))            # in order, every epoch once
    assert [s[1] for s in seen] == hist
    # N reported for an epoch is the count AFTER that epoch's events (train_gaussians.py logs after them)
    n, want_n = 100, []
    for e in range(epochs):
        if e in (10, 30):
            n *= 2
        if e in (30, 45):
            n -= n // 10
        if e == 20:
            n -= n // 10
        want_n.append(n)
    assert [s[2] for s in seen] == want_n
    # read-backs: at least one before every event epoch's events and one at the end; none in between beyond the cadence
    sync_epochs = [x[1] for x in tr.log if x[0] == "sync"]
    for e in (10, 20, 30, 45, epochs - 1):
        assert e in sync_epochs
    if sync_every >= 1000:  # capped at 60 epochs per read-back (the device parks at most 64 sums)
        assert sync_epochs == [10, 20, 30, 45, epochs - 1]
    if sync_every == 1:
        assert sync_epochs == list(range(epochs))


def test_calendar_strategies_and_regulariser_cadence(cfg):
    tr, hist, seen = _run(cfg, 8)
    steps = [x for x in tr.log if x[0] == "steps"]
    # every iteration is enqueued exactly once, in the caller's view order
    flat = [(x[1], v) for x in steps for v in x[2]]
    assert flat == [(e, (e + i) % 7) for e in range(60) for i in range(7)]
    # strategy: `loss_before_alternating` up to the start epoch, then less_freq on every `period`-th GLOBAL step
    pl = cfg["training"]["loss"]["projection_losses"]
    period = pl["sampling_whole_num_epochs_ratio"]
    k = 0
    for x in steps:
        for s in x[3]:
            e = x[1]
            if e > 5:
                assert s == (pl["less_freq_loss"] if k % period == 0 else pl["more_freq_loss"]), (e, k, s)
            else:
                assert s == pl["loss_before_alternating"]
            k += 1
    # regularisers: after epoch 40, direction then ratio after every 5th global step; the steps in between go out
    # as ONE native run
    regs = [x for x in tr.log if x[0] == "reg"]
    assert regs and all(r[2] % 5 == 0 for r in regs)
    first = min(r[2] for r in regs)
    assert first >= 41 * 7 and first < 41 * 7 + 5
    kinds = [r[1] for r in regs]
    assert kinds[0::2] == ["direction"] * (len(kinds) // 2) and kinds[1::2] == ["ratio"] * (len(kinds) // 2)
    late = [x for x in steps if x[1] > 41]
    assert max(len(x[2]) for x in late) <= 5 and max(len(x[2]) for x in steps if x[1] < 40) == 7
    # after an event: absgrads reset, rows re-sorted, capacity re-sized -- in that order, once per event epoch
    for e in (10, 20, 30, 45):
        tail = [x[0] for x in tr.log if len(x) > 1 and x[1] == e and x[0] in ("reset_absgrads", "sort")]
        assert tail == ["reset_absgrads", "sort"]


def test_unknown_dup_threshold_type_raises_like_the_reference(cfg):
    model_cfg, training_cfg = dict(cfg["model"]), cfg["training"]
    model_cfg.update(dup_high_pos_grads_at_epoch=[1], dup_threshold_type="percentile")
    with pytest.raises(NotImplementedError):
        train_loop.train(FakeTrainer(), model_cfg, training_cfg, lambda e: [0, 1], num_epochs=3)


def test_lr_schedule_is_installed_from_the_config(cfg):
    tr = FakeTrainer()
    train_loop.train(tr, dict(cfg["model"]), cfg["training"], lambda e: [0], num_epochs=1)
    assert isinstance(tr.schedule, LRSchedule)
    want = cfg["training"]["optim"]
    lr0, lr35 = tr.schedule.at(0), tr.schedule.at(35)
    assert lr0["means"] == pytest.approx(want["means"]["start_lr"])
    assert lr35["means"] == pytest.approx(want["means"]["start_lr"] * want["means"]["gamma"] ** 3)  # milestones 10 20 30
    assert lr0["opacities"] == 0.0 and lr35["opacities"] == pytest.approx(want["opacities"]["start_lr"])  # from epoch 20
    assert lr0["scales"] == 0.0 and lr35["scales"] == pytest.approx(want["scales"]["start_lr"])


def test_replica_calendar_matches_the_reference_driver(golden_dir):
    """configs/Replica.json through the reference's own `train()` (recorded by tests/golden/make_golden.py
    `replica_calendar`) against this driver: the same events after the same epochs -- including the absgrad reset of
    the `cull_wayward` epochs 45 and 300, whose mask the reference computes and never applies (edge_gs.py:498-542,
    train_gaussians.py:204-208,218-219) -- and the config's `reset_opacity_value` / cull value reach the culls."""
    ref = json.load(open(os.path.join(golden_dir, "replica_calendar.json")))
    name = {"duplicate_high_pos_gradients": "dup", "cull_gaussians_opacity": "cull_opacity",
            "cull_gaussians_not_projecting": "cull_np", "reset_absgrads": "reset_absgrads"}
    want = [(name[c], e) for e, c in ref["calls_after_epoch"] if c in name]
    assert ("reset_absgrads", 45) in want and ("reset_absgrads", 300) in want  # the wayward epochs
    tr = FakeTrainer()
    model_cfg = dict(ref["model"], reset_opacity_value=0.11)
    train_loop.train(tr, model_cfg, ref["training"], lambda e: [0, 1, 2], edge_masks_u8=object())
    got = [(x[0], x[1]) for x in tr.log if x[0] in name.values()]
    assert got == want
    culls = [x for x in tr.log if x[0] == "cull_opacity"]
    assert culls and all(x[2] == ref["model"]["cull_opacity_value"] and x[3] == 0.11 for x in culls)
    # a wayward epoch changes no rows: no re-sort, no capacity sweep
    assert not [x for x in tr.log if x[0] == "sort" and x[1] in (45, 300)]
    assert ref["if_reset_opacity_as_parsed"] is False  # "if reset_opacity" (configs/*.json:37) never reaches the dataclass
    with pytest.raises(NotImplementedError):
        train_loop.train(FakeTrainer(), dict(model_cfg, if_reset_opacity=True), ref["training"], lambda e: [0], num_epochs=1)


def test_on_epoch_defaults_to_the_reference_cadence(cfg):
    """Without an explicit `sync_every` a callback is called after ITS epoch (one read-back per epoch, like the
    reference's loop); without a callback the loop reads back every 8 epochs."""
    tr = FakeTrainer()
    train_loop.train(tr, dict(cfg["model"]), cfg["training"], lambda e: [0, 1], num_epochs=9,
                     on_epoch=lambda e, l, n: None)
    assert [x[1] for x in tr.log if x[0] == "sync"] == list(range(9))
    tr = FakeTrainer()
    train_loop.train(tr, dict(cfg["model"]), cfg["training"], lambda e: [0, 1], num_epochs=9)
    assert [x[1] for x in tr.log if x[0] == "sync"] == [7, 8]


class FakeDP:
    """A rank of a 2-way DataParallelStep over a FakeTrainer: records what the rank was asked to rasterise."""

    def __init__(self, tr, rank, world=2):
        self.worker, self.rank, self.world = tr, rank, world

    def step(self, view, wmap, next_view=None):
        views = [view] if isinstance(view, int) else list(view)
        wm = [wmap] if isinstance(view, int) else list(wmap)
        self.worker.log.append(("dp", self.worker.epoch, views, [w[0] for w in wm]))
        self.worker._journal.append(tuple(views))

    def reduce_words(self, ints, floats):  # (the other rank agrees with this one)
        return list(ints), list(floats)


def test_views_per_step_batches_and_shards_like_the_single_process_run(cfg):
    """C = 2 views per optimizer step: the single-process run issues batched steps over consecutive pairs of the epoch's
    order (a short last batch is filled from the front); under a 2-way data-parallel driver rank r rasterises view r of
    every pair with the SAME strategy the single process gives that view, skips the `bg_edge_ratio` draws of the
    other rank's views (the draw sequence stays in step), and the regulariser fires whenever a multiple of five VIEWS
    has been passed."""
    model_cfg, training_cfg = dict(cfg["model"]), json.loads(json.dumps(cfg["training"]))
    model_cfg.update(if_duplicate_high_pos_grad=False, if_cull_low_opacity=False, if_cull_gaussians_not_projecting=False)
    ol, pl = training_cfg["loss"]["orientation_losses"], training_cfg["loss"]["projection_losses"]
    ol["start_dir_loss_at_epoch"], ol["start_ratio_loss_at_epoch"] = 1, 99
    pl["start_alternating_at_epoch"] = -1
    order = lambda e: [(3 * e + i) % 7 for i in range(7)]  # noqa: E731  (7 views: the last pair is filled up)
    single = FakeTrainer()
    train_loop.train(single, model_cfg, training_cfg, order, num_epochs=3, views_per_step=2)
    batches = [x for x in single.log if x[0] == "batch"]
    assert [len(b[2]) for b in batches] == [2] * 12 and sum(len(b[2]) for b in batches) == 24
    assert batches[3][2] == [order(0)[6], order(0)[0]]  # the short batch wraps to the front of the epoch's order
    period = pl["sampling_whole_num_epochs_ratio"]
    flat = [s for b in batches for s in b[3]]
    assert flat == [pl["less_freq_loss"] if k % period == 0 else pl["more_freq_loss"] for k in range(24)]
    regs = [x for x in single.log if x[0] == "reg"]
    # direction regulariser from epoch 2 on (epoch > 1): views 16..24 pass the multiples 20 (after the batch 18-19 -> 20)
    assert [r[2] for r in regs] == [20] and all(r[1] == "direction" for r in regs)
    ranks = []
    for r in range(2):
        tr = FakeTrainer()
        train_loop.train(tr, model_cfg, training_cfg, order, num_epochs=3, views_per_step=2, dp=FakeDP(tr, r))
        ranks.append(tr)
        mine = [x for x in tr.log if x[0] == "dp"]
        assert [m[2] for m in mine] == [[b[2][r]] for b in batches]
        assert [m[3] for m in mine] == [[b[3][r]] for b in batches]
        # the other rank's bg_edge_ratio views: draws skipped, one per such view
        others = sum(1 for b in batches if b[3][1 - r] == "bg_edge_ratio")
        assert sum(1 for x in tr.log if x[0] == "skip_draw") == others
        # the regulariser fires on every rank after the same optimizer step as in the single-process run
        pos = lambda log, key: [sum(1 for y in log[:i] if y[0] == key) for i, x in enumerate(log) if x[0] == "reg"]  # noqa: E731
        assert pos(tr.log, "dp") == pos(single.log, "batch") == [10]
    with pytest.raises(ValueError):
        train_loop.train(FakeTrainer(), model_cfg, training_cfg, order, num_epochs=1, views_per_step=3,
                         dp=FakeDP(FakeTrainer(), 0))


class FakeDPOtherRankBig:
    """Two-rank data-parallel driver as train() sees it from rank 0; `other_is_big` plays the other rank's answer."""
    world, rank = 2, 0

    def __init__(self, tr, other_is_big):
        self.tr, self.other_is_big, self.asked = tr, other_is_big, 0

    def step(self, view, wmap):
        self.tr.train_step_batched([view] if isinstance(view, int) else view, [wmap] if isinstance(view, int) else wmap)

    def reduce_words(self, ints, floats):
        self.asked += 1
        return [max(i, int(self.other_is_big())) for i in ints], list(floats)


def test_read_back_decision_is_collective_under_data_parallelism(cfg):
    """the journal's size is rank-local (a rank journals only its own views' weight maps), the
    read-back is a collective.  If rank 1's journal crosses the 512 MB mark, rank 0 -- whose own journal is small --
    must read back in the same epoch: the ranks agree on the decision through dp.reduce_words every epoch."""
    model_cfg, training_cfg = dict(cfg["model"]), json.loads(json.dumps(cfg["training"]))
    model_cfg.update(dup_high_pos_grads_at_epoch=[], cull_opacity_at_epoch=[], cull_gaussians_not_projecting_at_epoch=[])
    for big_epochs in ({3, 4}, set()):
        tr = FakeTrainer()
        dp = FakeDPOtherRankBig(tr, lambda: tr.epoch in big_epochs)
        train_loop.train(tr, model_cfg, training_cfg, lambda e: [(e + i) % 6 for i in range(6)], edge_masks_u8=object(),
                         num_epochs=12, sync_every=1000, views_per_step=2, dp=dp)
        syncs = [x[1] for x in tr.log if x[0] == "sync"]
        assert dp.asked == 12, "the decision is agreed on once per epoch"
        assert syncs == (sorted(big_epochs) + [11] if big_epochs else [11]), syncs
