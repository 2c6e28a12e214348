"""EdgeTrainer -- the reference's per-view training step as ONE natively sequenced enqueue.

Host-side mirror of the hot loop ``train_epoch`` body, ``/root/reference/train_gaussians.py:71-106``
and of the model methods it drives (``edgegaussians/models/edge_gs.py``):

    output = model(idx)                       edge_gs.py:617-623,197-286
    loss = model.compute_projection_loss(..)  edge_gs.py:288-324   (weight-map form)
    (lambda * loss).backward()                train_gaussians.py:98,101
    model.update_absgrads()                   edge_gs.py:607-613
    for opt in 4 Adams: step(); zero_grad()   train_gaussians.py:104-106, train_utils.py:50-60

and of the epoch-boundary events: ``duplicate_high_pos_gradients`` (edge_gs.py:544-576),
``cull_gaussians_opacity`` (:477-488), ``cull_gaussians_not_projecting`` (:578-601).

Differences from the reference that are deliberate (documented in DESIGN.md):
  * GT images, weight maps and cameras are device-resident; no per-step H2D, no ``.item()``
    syncs (the loss accumulates in a device scalar that is read when the caller asks);
  * the RNG-dependent ``bg_edge_ratio`` sample mask is an INPUT (a per-pixel weight map built by
    the caller, e.g. ``synth.weight_map``), because the reference draws it with the CPU generator;
  * M (tile intersections) never crosses to the host inside a step: the isect buffers have a
    capacity sized from a count-only sweep over all views, and an overflow flag is checked lazily.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import AdamHyper, StepArgs, call, ptr, stream

TILE = 16


class IsectOverflow(RuntimeError):
    """A view produced more tile intersections than the isect buffers hold and the affected steps could
    not be replayed (journal off, or the data-parallel leg).  Results since the last check are invalid."""


@dataclass
class LRSchedule:
    """Per-epoch learning rates of ``train_utils.get_optimizers_schedulers`` (train_utils.py:48-65):
    MultiStepLR for means, 'zero until start_at_epoch, then constant' for the other three."""
    means_lr: float = 2e-3
    means_milestones: List[int] = field(default_factory=lambda: [10, 20, 30, 40, 50])
    means_gamma: float = 0.75
    scales_lr: float = 1e-4
    scales_start: int = 30
    quats_lr: float = 1e-3
    quats_start: int = 30
    opacities_lr: float = 0.03
    opacities_start: int = 20

    @classmethod
    def from_config(cls, optim_cfg: Dict) -> "LRSchedule":
        m, s, q, o = optim_cfg["means"], optim_cfg["scales"], optim_cfg["quats"], optim_cfg["opacities"]
        return cls(m["start_lr"], list(m["milestones"]), m["gamma"], s["start_lr"], s["start_at_epoch"],
                   q["start_lr"], q["start_at_epoch"], o["start_lr"], o["start_at_epoch"])

    def at(self, epoch: int) -> Dict[str, float]:
        k = sum(1 for ms in self.means_milestones if ms <= epoch)
        return {
            "means": self.means_lr * (self.means_gamma ** k),
            "scales": 0.0 if epoch < self.scales_start else self.scales_lr,
            "quats": 0.0 if epoch < self.quats_start else self.quats_lr,
            "opacities": 0.0 if epoch < self.opacities_start else self.opacities_lr,
        }


class EdgeTrainer:
    """Device-resident training state + fused step.  One instance per GPU (one process per GPU)."""

    def __init__(self, means: Tensor, log_scales: Tensor, quats: Tensor, logit_opacities: Tensor,
                 viewmats: Tensor, Ks: Tensor, gt: Tensor, width: int, height: int,
                 device: str = "cuda", schedule: Optional[LRSchedule] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8, keep_images: bool = False,
                 spatial_order: bool = False, segmented: Optional[bool] = None, replay_on_overflow: bool = True,
                 seed: int = 0):
        _lib.load()
        # replay_on_overflow: the steps enqueued since the last read-back (pop_loss) are journalled and the
        # state before them is kept, so that an intersection overflow -- noticed at the read-back through
        # the STICKY device flag total[1] -- is repaired by growing the buffers and re-running exactly those
        # steps instead of silently training on dropped intersections.  Costs one state copy (136 B per
        # Gaussian) per run of steps between read-backs.
        self.replay_on_overflow = bool(replay_on_overflow)
        self._journal: List = []
        self._snap: Optional[Dict] = None
        self.overflow_events = 0
        self.rewalk_misses = 0  # replays caused by a transmittance stop while the re-walk launch was being skipped
        self.rewalk_hint = -1  # re-walk list length seen at the last read-back (launch-shape hint; -1 = unknown)
        self._projected: Optional[int] = None  # view already projected + binned by apply_adam(next_view=...)
        self._dp = None  # the DataParallelStep driving this trainer (dist.py), if any: read-backs and replays are collective
        self._ws_tag = 0       # tags of the chained forward (eg_step_args.ws_tag): one fresh value per enqueued step
        self._replaying = False  # inside _recover_from_overflow's replay of the journal
        self.chained_forward = bool(int(os.environ.get("EG_CHAINED", "1")))
        # round 6 (eg_step_args.two_kernel_backward): inside a native run of steps on a tile grid of <= 2048 tiles the backward of
        # a scene of <= 32768 Gaussians is ONE kernel (csrc/backward_fused.hip); != 0 (development: EG_TWO_KERNEL_BACKWARD=1)
        # keeps the two kernels of rounds 1-5 -- the same parameters bit for bit (tests/test_gpu_parity.py)
        self.two_kernel_backward = int(os.environ.get("EG_TWO_KERNEL_BACKWARD", "0"))  # eg_step_args.two_kernel_backward
        # noise of duplicate() comes from a dedicated generator seeded with (seed, event number): identical
        # on every data-parallel rank whatever else the ranks drew (edge_gs.py:462-467 uses the global RNG)
        self.seed = int(seed)
        self._dup_events = 0
        # keep_images: also materialise render / alphas / last_ids / vpix every step (the training step
        # itself needs none of them: its backward reads only the packed gtstop record)
        self.keep_images = bool(keep_images)
        # segmented (default; EG_SEGMENTED=0 or segmented=False selects the count / scan / emit sequence of
        # the operator path): the training step bins with fixed per-tile key segments -- projection +
        # binning in one kernel, no emit pass; same results, the isect buffers become [T * seg_cap]
        self.segmented = bool(int(os.environ.get("EG_SEGMENTED", "1"))) if segmented is None else bool(segmented)
        self.seg_cap = 0
        self.dev = torch.device(device)
        f = dict(device=self.dev, dtype=torch.float32)
        self.means = means.detach().to(**f).contiguous().clone()
        self.log_scales = log_scales.detach().to(**f).contiguous().clone()
        self.quats = quats.detach().to(**f).contiguous().clone()
        self.logit_opacities = logit_opacities.detach().to(**f).reshape(-1).contiguous().clone()
        self.viewmats = viewmats.to(**f).contiguous()
        self.Ks = Ks.to(**f).contiguous()
        self.gt = gt.to(**f).contiguous()  # [V,H,W] in [0,1]
        self.width, self.height = int(width), int(height)
        self.tw, self.th = math.ceil(width / TILE), math.ceil(height / TILE)
        self.T = self.tw * self.th
        self.V = self.viewmats.shape[0]
        self.schedule = schedule or LRSchedule()
        self.betas, self.eps = betas, eps
        self.adam_step = 0  # step count of the opacity optimizer (== all four until a regulariser steps)
        self.group_steps = [0, 0, 0, 0]  # per-optimizer counts: means, scales, quats, opacities
        self.step = 0       # model.step (edge_gs.py:621)
        self.epoch = 0
        self.loss_scale = 1.0  # lambda_projection (train_gaussians.py:98; constant 1 in every config)
        self.capacity = 0
        self._hyper = AdamHyper()
        self._alloc_state()
        self._alloc_pixels()
        # spatial_order: keep the Gaussian rows in 3-D Morton order (re-sorted after every densify / cull
        # event).  Neighbouring rows then project to neighbouring pixels in every view, so a binning
        # workgroup touches a handful of tile counters instead of hundreds and the gathers of the
        # compositing kernels stay local.  ref_index remembers where each row sits in the reference's
        # own arrays; state_dict() / export_as_ply() hand the rows back in that order.
        self.spatial_order = bool(spatial_order)
        self.ref_index: Optional[Tensor] = None
        if self.spatial_order:
            self.spatial_sort()

    # ------------------------------------------------------------------ allocation
    @property
    def N(self) -> int:
        return self.means.shape[0]

    def _alloc_state(self):
        N, d = self.N, self.dev
        self.adam_m = torch.zeros(11 * N, device=d)
        self.adam_v = torch.zeros(11 * N, device=d)
        self.absgrads = torch.zeros(N, device=d)
        self.absgrads_normalize_factor = 1.0  # edge_gs.py:89,605
        self._alloc_per_gaussian()

    def _alloc_per_gaussian(self):
        self._drop_projection()
        N, d = self.N, self.dev
        self.splat = torch.empty(N, 8, device=d)
        self.__dict__.pop("_knn_buf", None)  # (neighbour table + per-row K-th distances: the rows have changed)
        self.g2d = torch.empty(N, 8, device=d)  # written (not accumulated) by the footprint backward
        self.tile_mask = torch.zeros(N, dtype=torch.int32, device=d)  # exact tile hits per Gaussian (bit mask)
        self.grads = torch.zeros(N, 12, device=d)  # [means3|quats4|scales3|opac1|absgrad-inc1] for all-reduce
        self._grads_b = None
        self._args_cache: Dict = {}
        self._batches: Dict = {}  # C -> [C, ...] work buffers of train_step_batched (allocated on first use)
        self._grads_b = None      # second gradient buffer (data-parallel half-step overlap)

    def _alloc_pixels(self):
        H, W, d = self.height, self.width, self.dev
        self.tile_counts = torch.zeros(self.T, dtype=torch.int32, device=d)
        self.offsets = torch.zeros(self.T + 1, dtype=torch.int32, device=d)
        self.item_offsets = torch.zeros(self.T + 1, dtype=torch.int32, device=d)
        self.total = torch.zeros(4, dtype=torch.int32, device=d)  # M, overflow, items, largest tile
        # [0]: last-workgroup ticket of the fused scan; [1 .. T + 1]: its front-slice prefix (EG_FLAG_FRONT_PREFIX)
        self.ticket = torch.zeros(self.T + 2, dtype=torch.int32, device=d)
        img = (lambda **k: torch.zeros(H, W, device=d, **k)) if self.keep_images else (lambda **k: None)
        self.render, self.alphas, self.vpix = img(), img(), img()
        # {vpix * T_final, stop id, stop depth bits} for the fused backward, initialised to "nothing contributed, no stop"
        # {0, -1, all ones}: on grids of <= 2048 tiles the fused forward does not write the pixels of EMPTY tiles (no
        # footprint reaches them; include/edgegs.h, item_rec), which keep what they held
        self.gtstop = torch.zeros(H, W, 3, device=d)
        self.gtstop.view(torch.int32)[..., 1:] = -1
        self.last_ids = img(dtype=torch.int32)
        # running projection-loss sum (what the step kernels add to) + the parked sums of the epochs marked since the
        # last read-back (mark_epoch); one buffer: one device->host copy reads them all
        self._loss_buf = torch.zeros(1 + 64, device=d)
        self.loss_acc = self._loss_buf[:1]
        self._n_marks = 0

    def _alloc_isect(self, capacity: int, seg_cap: int = 0):
        """capacity: upper bound on the tile intersections of one view (sizes the item workspace);
        seg_cap > 0: segmented binning, the key / id arrays hold T fixed segments of seg_cap slots."""
        self._drop_projection()
        self.capacity = int(capacity)
        self.seg_cap = int(seg_cap)
        n_keys = max(self.T * self.seg_cap, self.capacity)  # (the staged path always uses the classic layout)
        self.keys = torch.empty(n_keys, dtype=torch.int64, device=self.dev)
        self.flatten_ids = torch.empty(n_keys, dtype=torch.int32, device=self.dev)
        self.max_items = max((self.capacity + 127) // 128 + self.T, getattr(self, "_rec_need", 0))
        if self.seg_cap:
            self.tile_end = torch.zeros(self.T, dtype=torch.int32, device=self.dev)
            self.item_end = torch.zeros(self.T, dtype=torch.int32, device=self.dev)
            self.item_tile = torch.zeros(self.max_items, dtype=torch.int32, device=self.dev)
            # the sort kernel's per-item records for the wave-autonomous forward (composite_wave.hip)
            self.item_rec = torch.zeros(self.max_items, 4, dtype=torch.int32, device=self.dev)
        self.workspace = _lib.composite_workspace(self.max_items, self.T, self.dev)
        self._args_cache = {}
        self._batches = {}

    # ------------------------------------------------------------------ loss weight maps
    def weight_map(self, view: int, strategy: str, ratio: float = 1.0,
                   generator: Optional[torch.Generator] = None, threshold: float = 0.5) -> Tensor:
        """Per-pixel weights w with  loss = sum_p w_p |render_p - gt_p|  for the reference's three
        strategies (edge_gs.py:288-324).  'whole' / 'weighted' are cached on the device; the
        'bg_edge_ratio' sample is drawn per call -- on the device by default, or on the host from
        `generator` exactly like the reference (CPU randperm, edge_gs.py:306) when one is given."""
        from . import synth
        cache = self.__dict__.setdefault("_wmaps", {})
        H, W = self.height, self.width
        hw = H * W
        if strategy == "whole":
            if "whole" not in cache:
                cache["whole"] = torch.full((H, W), 1.0 / hw, device=self.dev)
            return cache["whole"]
        key = ("edge", view, threshold)
        if key not in cache:
            edge = (self.gt[view] >= threshold)
            cache[key] = (edge, int(edge.sum().item()))
        edge, n_e = cache[key]
        if strategy == "weighted":
            k2 = ("weighted", view, threshold)
            if k2 not in cache:
                n_b = hw - n_e
                cache[k2] = (torch.where(edge, n_b / hw, n_e / hw).float() / hw).contiguous()
            return cache[k2]
        if strategy == "bg_edge_ratio":
            if generator is not None:
                return synth.weight_map(strategy, self.gt[view].cpu(), ratio, generator, threshold).to(self.dev)
            n_sel = int(ratio * n_e)
            # one launch: the sample (exactly n_sel distinct pixels) is drawn inside the kernel from a keyed
            # pseudo-random permutation; the key advances with every draw
            self._wmap_draws = getattr(self, "_wmap_draws", 0) + 1
            key = ((self.seed + 1) * 0x9E3779B97F4A7C15 + self._wmap_draws * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            w = torch.empty(H, W, device=self.dev)
            call("eg_ratio_wmap_seeded", ptr(self.gt[view]), float(threshold), n_e, hw - n_e, min(n_sel, hw - n_e), key,
                 hw, ptr(w), stream())
            return w
        raise ValueError(f"Unknown projection loss strategy: {strategy}")

    def weight_maps(self, views: List[int], strategies: List[str], ratio: float = 1.0, threshold: float = 0.5) -> List[Tensor]:
        """`[weight_map(v, s, ratio) for v, s in zip(views, strategies)]` with every device-side `bg_edge_ratio` draw of the
        list made by ONE native call (`eg_ratio_wmaps_seeded`) into one [C, H, W] block: the same maps, the same draw
        sequence -- what a run of steps needs in front of its enqueue (round 6: 13 us of host time per draw otherwise)."""
        out: List[Optional[Tensor]] = [None] * len(views)
        draws = []
        cache = self.__dict__.setdefault("_wmaps", {})
        hw = self.height * self.width
        for i, (v, st) in enumerate(zip(views, strategies)):
            if st != "bg_edge_ratio":
                out[i] = self.weight_map(v, st, ratio, None, threshold)
                continue
            key = ("edge", v, threshold)
            if key not in cache:
                edge = (self.gt[v] >= threshold)
                cache[key] = (edge, int(edge.sum().item()))
            n_e = cache[key][1]
            self._wmap_draws = getattr(self, "_wmap_draws", 0) + 1
            seed = ((self.seed + 1) * 0x9E3779B97F4A7C15 + self._wmap_draws * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            draws.append((i, v, n_e, hw - n_e, min(int(ratio * n_e), hw - n_e), seed))
        if draws:
            n = len(draws)
            block = torch.empty(n, self.height, self.width, device=self.dev)
            g0 = self.gt.data_ptr()
            call("eg_ratio_wmaps_seeded", n, (C.c_void_p * n)(*[g0 + 4 * hw * d[1] for d in draws]), float(threshold),
                 (C.c_int32 * n)(*[d[2] for d in draws]), (C.c_int32 * n)(*[d[3] for d in draws]),
                 (C.c_int32 * n)(*[d[4] for d in draws]), (C.c_uint64 * n)(*[d[5] for d in draws]), hw, ptr(block), stream())
            for k, d in enumerate(draws):
                out[d[0]] = block[k]
        return out

    def skip_weight_map_draw(self) -> None:
        """Advance the key of the device-side `bg_edge_ratio` draws without drawing: a data-parallel rank keeps its
        draw sequence in step with the single-process run (and with the other ranks) for the views it does not own."""
        self._wmap_draws = getattr(self, "_wmap_draws", 0) + 1

    # ------------------------------------------------------------------ capacity
    def count_intersections(self, view: int) -> int:
        """M for one view (count-only pass: projection + per-tile counts + scan).  Host sync."""
        self._drop_projection()  # (overwrites splat and the cursors a pre-projected view would still need)
        call("eg_project_fwd", ptr(self.means), ptr(self.quats), ptr(self.log_scales), ptr(self.logit_opacities),
             ptr(self.viewmats[view]), ptr(self.Ks[view]), self.N, self.width, self.height, 0.01, 1e10, 0.3, 0.0,
             _lib.FLAG_LOG_SCALES | _lib.FLAG_LOGIT_OPACITIES | _lib.FLAG_ANTIALIASED | _lib.FLAG_TIGHT_TILES,
             ptr(self.splat),
             None, None, None, None, None, None, ptr(self.tile_counts), None, stream())
        call("eg_tile_offsets", ptr(self.tile_counts), self.T, 1 << 40, ptr(self.offsets), ptr(self.item_offsets),
             ptr(self.total), stream())
        tot = self.total.tolist()  # one host sync: M, overflow flag, items, largest tile population
        self._last_tile_max = int(tot[3])
        self._last_rec_span = self._record_span()
        self.tile_counts.zero_()
        return int(tot[0])

    def _record_span(self) -> int:
        """Entries of the item-record table the view just counted needs (tile_counts holds its populations): the number
        of items -- or, with the XCD-aware placement (include/edgegs.h, item_rec), 8 x the longest of the eight per-XCD lists."""
        shift = int(_lib.load().eg_record_xcd_shift(self.T)) if self.segmented else 0
        if shift == 0:
            return 0  # (dense records: one per item, and capacity / 128 + T bounds the items)
        items = torch.clamp((self.tile_counts[:self.T] + 127) // 128, min=1)
        if getattr(self, "_tile_xcd", None) is None or self._tile_xcd[1] != shift:
            tw = (self.width + 15) // 16
            t = torch.arange(self.T, device=self.dev)
            if self.T <= _lib.PREFIX_HERE_MAX_TILES:  # 2 x 2-tile blocks dealt round-robin
                self._tile_xcd = ((((t % tw) >> shift) + 3 * ((t // tw) >> shift)) % 8, shift)
            else:                                      # (round 6) above 2048 tiles: bands of 2^shift tile rows
                self._tile_xcd = (((t // tw) >> shift) % 8, shift)
        per = torch.bincount(self._tile_xcd[0], weights=items.double(), minlength=8)
        return 8 * int(per.max().item())

    def ensure_capacity(self, slack: float = 1.3, views: Optional[List[int]] = None) -> int:
        """Sizes the isect buffers from a count-only sweep (called at start and after every
        densify / cull event, i.e. whenever N changes -- 22 times in a 400-epoch ABC run)."""
        self._drop_projection()
        views = list(range(self.V)) if views is None else views
        m_max, tile_max, span_max = 0, 0, 0
        for v in views:
            m_max = max(m_max, self.count_intersections(v))
            tile_max = max(tile_max, self._last_tile_max)
            span_max = max(span_max, self._last_rec_span)
        self.max_tile_seen = tile_max  # launch-shape hint of the tile sort (never affects results)
        # the item-record table (XCD-aware placement: 8 x the longest per-XCD list) with the same slack as the keys
        self._rec_need = int(span_max * slack) + 64 if span_max > 0 else 0
        need = int(m_max * slack) + 4096
        seg = (int(tile_max * 1.5) // 128 + 2) * 128 if self.segmented else 0
        if self.T * seg > (1 << 28):  # a few monster tiles would make T fixed segments absurdly large (> 3 GB):
            seg = 0                   # fall back to the count / scan / emit layout for this scene
            self.seg_cap = 0
        if need > self.capacity or seg > self.seg_cap or self._rec_need > self.max_items:
            self._alloc_isect(max(need, self.capacity), max(seg, self.seg_cap))
        self.m_max_seen = m_max
        self._args_cache = {}
        return m_max

    # ------------------------------------------------------------------ the step
    def _args(self, view: int, wmap: Tensor, fused_adam: bool, n_tags: int = 1) -> StepArgs:
        a = self._args_cache.get("sa")
        if a is None:  # rebuilt only when a buffer was re-allocated (N or capacity changed)
            a = StepArgs()
            a.means, a.quats = ptr(self.means), ptr(self.quats)
            a.log_scales, a.logit_opacities = ptr(self.log_scales), ptr(self.logit_opacities)
            a.adam_m, a.adam_v = ptr(self.adam_m), ptr(self.adam_v)
            a.N = self.N
            a.width, a.height = self.width, self.height
            a.splat, a.g2d = ptr(self.splat), ptr(self.g2d)
            a.gtstop = ptr(self.gtstop)
            a.max_tile_hint = getattr(self, "max_tile_seen", 0)
            a.seg_cap = self.seg_cap
            if self.seg_cap:
                a.tile_end, a.item_end, a.item_tile = ptr(self.tile_end), ptr(self.item_end), ptr(self.item_tile)
                a.item_rec = ptr(self.item_rec)
            a.tile_counts, a.offsets, a.total = ptr(self.tile_counts), ptr(self.offsets), ptr(self.total)
            a.item_offsets, a.workspace, a.max_items = ptr(self.item_offsets), ptr(self.workspace), self.max_items
            a.tile_mask, a.ticket = ptr(self.tile_mask), ptr(self.ticket)
            a.keys, a.flatten_ids, a.capacity = ptr(self.keys), ptr(self.flatten_ids), self.capacity
            a.render, a.alphas, a.vpix = ptr(self.render), ptr(self.alphas), ptr(self.vpix)
            a.loss, a.last_ids = ptr(self.loss_acc), ptr(self.last_ids)
            g0 = self.grads.data_ptr()
            a.v_means, a.v_quats = g0, g0 + 4 * 3 * self.N
            a.v_scales, a.v_opacities = g0 + 4 * 7 * self.N, g0 + 4 * 10 * self.N
            self._args_cache["sa"] = a
            self._args_cache["hyper_ptr"] = C.pointer(self._hyper)
            self._args_cache["null_hyper"] = C.POINTER(AdamHyper)()
        assert wmap.is_cuda and wmap.is_contiguous() and wmap.shape == (self.height, self.width)
        a.viewmat = self.viewmats.data_ptr() + 64 * view
        a.K = self.Ks.data_ptr() + 36 * view
        a.gt = self.gt.data_ptr() + 4 * self.height * self.width * view
        a.wmap = wmap.data_ptr()
        a.loss_scale = self.loss_scale
        a.rewalk_hint = self._rewalk_arg(fused_adam)
        a.ws_tag = self._next_tag(n_tags) if self.chained_forward else 0
        a.two_kernel_backward = 1 if (self.two_kernel_backward or getattr(self, "_side_by_side", False)) else 0
        if fused_adam:
            a.absgrads = ptr(self.absgrads)
            a.adam_host = self._args_cache["hyper_ptr"]
        else:  # data-parallel: the absgrad increment of this view goes to block 11 of `grads`
            a.absgrads = self.grads.data_ptr() + 4 * 11 * self.N
            a.adam_host = self._args_cache["null_hyper"]
        return a

    def attach_dp(self, dp) -> None:
        """Called by dist.DataParallelStep: from now on every read-back merges the sticky flags (max) and the loss sums
        (sum) over the ranks, and the journalled data-parallel steps are replayed by ALL ranks together."""
        self._dp = dp

    def _rewalk_arg(self, journalled: bool) -> int:
        """The re-walk launch hint of the next enqueue.  While no pixel has reached the transmittance stop (the
        whole phase before opacities train up, edge_gs.py:93 starts them at 0.08) the launch is skipped
        altogether -- speculation, covered by the journal: a stop raises a sticky device word, the read-back
        restores the state and replays the steps with the re-walk on."""
        if journalled and self.replay_on_overflow and self.rewalk_hint == 0:
            return _lib.REWALK_SPECULATE
        return self.rewalk_hint

    def _reserve_tags(self, n: int) -> None:
        """Make sure n fresh, consecutive tags are left in 1 .. EG_MAX_WS_TAG (16 bits: the forward's hand-over granules
        carry them).  When the range is used up (every 65 534 steps) the journal is flushed -- the sticky words of the
        control block are about to go: look at them first -- the workspaces are zeroed and the tags start over, so that
        a granule written 2^16 steps ago can never be mistaken for this call's.  Called BEFORE the steps are journalled
        and before their arguments are built.  (A replay reserves the tags of its whole journal before it starts,
        _recover_from_overflow: inside one this function never wraps.)"""
        assert 0 < n <= _lib.MAX_WS_TAG, n
        # (a replay draws up to two fresh tags per journalled entry and reserves them all up front: the journal is read
        # back -- flushed -- before it grows past half the tag range, whatever the forward's mode)
        if self._journal and not self._replaying and 2 * (len(self._journal) + n) > _lib.MAX_WS_TAG:
            self.flush()
        if self.chained_forward and self._ws_tag + n > _lib.MAX_WS_TAG:
            assert not self._replaying, "a replay reserves its tags up front"
            if self._journal:
                self.flush()
            else:  # nothing to replay, but the stall bit of control word 3 must not be zeroed unread
                if self._ctl_bits()[1]:
                    raise RuntimeError(self._STALL_MSG)
            self._zero_workspaces()

    def _zero_workspaces(self) -> None:
        for ws in [self.workspace] + [b["workspace"] for b in self._batches.values()]:
            if ws is not None:
                ws.zero_()
        # (the item records carry the call tag too: a record of 2^16 steps ago must not look like this call's)
        for rec in [getattr(self, "item_rec", None)] + [b.get("item_rec") for b in self._batches.values()]:
            if rec is not None:
                rec.zero_()
        self._ws_tag = 0

    def _next_tag(self, n: int) -> int:
        """First of n fresh, consecutive tags (see _reserve_tags, which the public entry points call before they
        journal; here it only catches a caller that did not)."""
        self._reserve_tags(n)
        t = self._ws_tag + 1
        self._ws_tag += n
        return t

    def _drop_projection(self) -> None:
        """Forget the view pre-projected by apply_adam(next_view=...).  Its binning has counted the tile cursors
        up and no sort has taken them back down: zero them, so that the next projection starts clean."""
        if getattr(self, "_projected", None) is not None and getattr(self, "tile_counts", None) is not None:
            self.tile_counts.zero_()
            self.ticket.zero_()
        self._projected = None

    def _ctl_words(self):
        """[(max re-walk list length, missed-re-walk flag)] of every compositing workspace in use (one small D2H each)."""
        o = 4 * (self.T + self.max_items + 2)
        out = [self.workspace[o:o + 8].view(torch.int32)]
        for b in self._batches.values():
            out.append(b["workspace"][:, o:o + 8].contiguous().view(torch.int32))
        return out

    _STALL_MSG = ("composite forward: a look-back poll gave up (a hand-over granule never arrived); "
                  "the results of the steps since the last read-back are invalid")

    def _ctl_bits(self):
        """Control word 3 of the compositing workspaces, as (missed, stalled): bit 0 = a pixel reached the transmittance
        stop while the forward speculated that none would (replay in chained mode); bit 1 = a wave of the forward gave
        up polling a hand-over granule -- the dispatch-order contract of the look-back was broken (results are void), or
        the view overflowed its item table and the forward ran on a truncated one (the overflow handling repairs it)."""
        words = [int(x) for w in self._ctl_words() for x in w.view(-1, 2)[:, 1].reshape(-1).tolist()]
        return any(x & 1 for x in words), any(x & 2 for x in words)

    def _advance_all(self):
        self.adam_step += 1
        self.group_steps = [t + 1 for t in self.group_steps]

    def _set_hyper(self):
        lr = self.schedule.at(self.epoch)
        h = self._hyper
        h.lr_means, h.lr_scales, h.lr_quats, h.lr_opacities = lr["means"], lr["scales"], lr["quats"], lr["opacities"]
        h.beta1, h.beta2, h.eps = self.betas[0], self.betas[1], self.eps
        h.step = max(self.adam_step, 1)
        for i in range(4):
            h.group_steps[i] = self.group_steps[i]

    def train_step(self, view: int, wmap: Tensor) -> None:
        """One reference iteration (train_gaussians.py:81-106) for `view`, fully asynchronous.
        `wmap` [H,W]: the per-pixel loss weights of the strategy chosen for this step."""
        if self.capacity == 0:
            self.ensure_capacity()
        self._reserve_tags(1)
        if self.replay_on_overflow:
            if not self._journal:
                self._snapshot()
            self._journal.append(("1", view, wmap, self.epoch, self.loss_scale))
        self._step_raw(view, wmap)

    def train_steps(self, views: List[int], wmaps: List[Tensor]) -> None:
        """len(views) consecutive reference iterations (one optimizer step per view, exactly `train_step` in a loop)
        enqueued by ONE native call: no Python between the steps.  The learning rates / loss scale of the current
        epoch apply to all of them."""
        K = len(views)
        if K == 0:
            return
        if self.capacity == 0:
            self.ensure_capacity()
        self._reserve_tags(K)
        if self.replay_on_overflow:
            if not self._journal:
                self._snapshot()
            self._journal.extend(("1", v, w, self.epoch, self.loss_scale) for v, w in zip(views, wmaps))
        self._steps_raw(views, wmaps)

    def _steps_raw(self, views, wmaps) -> None:
        a, va, wa = self._steps_begin(views, wmaps)
        call("eg_train_steps", C.byref(a), len(views), va, wa, ptr(self.viewmats), ptr(self.Ks), ptr(self.gt), stream())
        self._steps_end(len(views))

    def _steps_begin(self, views, wmaps):
        """Host state of a native run of len(views) steps: (argument block of step 0, view indices, weight-map pointers)."""
        self._drop_projection()
        K = len(views)
        self._advance_all()   # step 0's counts; the native loop advances them by k
        self._set_hyper()
        a = self._args(views[0], wmaps[0], True, n_tags=K)  # (the native loop uses ws_tag .. ws_tag + K - 1)
        va = (C.c_int32 * K)(*views)
        wa = (C.c_void_p * K)(*[w.data_ptr() for w in wmaps])
        for w in wmaps:
            assert w.is_cuda and w.is_contiguous() and w.shape == (self.height, self.width)
        return a, va, wa

    def _steps_end(self, K: int) -> None:
        for _ in range(K - 1):
            self._advance_all()
        self.absgrads_normalize_factor += K
        self.step += K

    def _step_raw(self, view: int, wmap: Tensor) -> None:
        self._drop_projection()
        self._advance_all()
        self._set_hyper()
        call("eg_train_step", C.byref(self._args(view, wmap, True)), stream())
        self.absgrads_normalize_factor += 1  # edge_gs.py:613
        self.step += 1

    # ------------------------------------------------------------------ C views per launch sequence
    def _alloc_batch(self, Cn: int):
        N, T, d = self.N, self.T, self.dev
        lib = _lib.load()
        stride = int(lib.eg_batched_workspace_stride(self.max_items, T))
        ctl = int(lib.eg_composite_workspace_ctl_bytes(self.max_items, T))
        ws = torch.zeros(Cn, stride, dtype=torch.uint8, device=d)  # (all of it: granule tag 0 = never written)
        i32 = dict(dtype=torch.int32, device=d)
        b = dict(C=Cn, splat=torch.empty(Cn, N, 8, device=d), g2d=torch.empty(Cn, N, 8, device=d),
                 tile_counts=torch.zeros(Cn, T, **i32), offsets=torch.zeros(Cn, T, **i32),
                 tile_end=torch.zeros(Cn, T, **i32), item_offsets=torch.zeros(Cn, T, **i32),
                 item_end=torch.zeros(Cn, T, **i32), item_tile=torch.zeros(Cn, self.max_items, **i32),
                 keys=torch.empty(Cn, T * self.seg_cap, dtype=torch.int64, device=d),
                 flatten_ids=torch.empty(Cn, T * self.seg_cap, **i32), total=torch.zeros(Cn, 4, **i32),
                 ticket=torch.zeros(Cn, **i32), gtstop=torch.zeros(Cn, self.height, self.width, 3, device=d),
                 workspace=ws, ws_stride=stride, rewalk_hint=-1)
        b["item_rec"] = torch.zeros(Cn, self.max_items, 4, **i32)
        b["gtstop"].view(torch.int32)[..., 1:] = -1  # (the neutral record: see _alloc_pixels)
        a = StepArgs()
        a.means, a.quats = ptr(self.means), ptr(self.quats)
        a.log_scales, a.logit_opacities = ptr(self.log_scales), ptr(self.logit_opacities)
        a.adam_m, a.adam_v = ptr(self.adam_m), ptr(self.adam_v)
        a.N, a.width, a.height = N, self.width, self.height
        a.splat, a.g2d, a.gtstop = ptr(b["splat"]), ptr(b["g2d"]), ptr(b["gtstop"])
        a.seg_cap, a.max_items, a.capacity = self.seg_cap, self.max_items, self.capacity
        a.tile_counts, a.offsets, a.total = ptr(b["tile_counts"]), ptr(b["offsets"]), ptr(b["total"])
        a.tile_end, a.item_end, a.item_tile = ptr(b["tile_end"]), ptr(b["item_end"]), ptr(b["item_tile"])
        a.item_offsets, a.workspace, a.ticket = ptr(b["item_offsets"]), ptr(ws), ptr(b["ticket"])
        a.keys, a.flatten_ids, a.loss = ptr(b["keys"]), ptr(b["flatten_ids"]), ptr(self.loss_acc)
        a.item_rec = ptr(b["item_rec"])
        b["args"] = a
        b["ptrs"] = [(C.c_void_p * Cn)() for _ in range(4)]
        b["hyper_ptr"] = C.pointer(self._hyper)
        b["null_hyper"] = C.POINTER(AdamHyper)()
        self._batches[Cn] = b
        return b

    def _batched_raw(self, views, wmaps, fused_adam: bool, slot: int = 0, journalled: Optional[bool] = None) -> Tensor:
        self._drop_projection()
        Cn = len(views)
        if not self.seg_cap:
            raise RuntimeError("train_step_batched needs the segmented binning layout (EdgeTrainer(segmented=True))")
        b = self._batches.get(Cn) or self._alloc_batch(Cn)
        a = b["args"]
        if slot and self._grads_b is None:
            self._grads_b = torch.zeros_like(self.grads)
        out = self._grads_b if slot else self.grads
        g0, N = out.data_ptr(), self.N
        a.v_means, a.v_quats = g0, g0 + 4 * 3 * N
        a.v_scales, a.v_opacities = g0 + 4 * 7 * N, g0 + 4 * 10 * N
        vm, Kp, gp, wp = b["ptrs"]
        hw4 = 4 * self.height * self.width
        for i, (v, w) in enumerate(zip(views, wmaps)):
            assert w.is_cuda and w.is_contiguous() and w.shape == (self.height, self.width)
            vm[i] = self.viewmats.data_ptr() + 64 * v
            Kp[i] = self.Ks.data_ptr() + 36 * v
            gp[i] = self.gt.data_ptr() + hw4 * v
            wp[i] = w.data_ptr()
        a.loss_scale = self.loss_scale
        a.max_tile_hint = getattr(self, "max_tile_seen", 0)
        a.ws_tag = self._next_tag(1) if self.chained_forward else 0
        journalled = fused_adam if journalled is None else journalled
        a.rewalk_hint = (_lib.REWALK_SPECULATE if (journalled and self.replay_on_overflow and b["rewalk_hint"] == 0
                                                   and self.rewalk_hint in (0, -1)) else b["rewalk_hint"])
        if fused_adam:
            self._advance_all()
            self._set_hyper()
            a.absgrads, a.adam_host = ptr(self.absgrads), b["hyper_ptr"]
        else:
            a.absgrads, a.adam_host = g0 + 4 * 11 * N, b["null_hyper"]
        call("eg_train_step_batched", C.byref(a), Cn, vm, Kp, gp, wp, stream())
        if fused_adam:
            self.absgrads_normalize_factor += Cn  # C calls of update_absgrads (edge_gs.py:613)
        self.step += Cn
        return out

    def train_step_batched(self, views: List[int], wmaps: List[Tensor]) -> None:
        """C = len(views) <= 8 views in ONE launch sequence (gridDim.y = view) and ONE optimizer step on the SUM of
        their gradients -- the semantics of C-way data parallelism (dist.py) on a single GPU.  A throughput mode:
        the reference steps after every view (train_gaussians.py:104-106)."""
        if self.capacity == 0:
            self.ensure_capacity()
        views, wmaps = list(views), list(wmaps)
        self._reserve_tags(1)
        if self.replay_on_overflow:
            if not self._journal:
                self._snapshot()
            self._journal.append(("b", views, wmaps, self.epoch, self.loss_scale))
        self._batched_raw(views, wmaps, True)

    def grad_step_batched(self, views: List[int], wmaps: List[Tensor], slot: int = 0, journalled: bool = False) -> Tensor:
        """Forward + loss + backward of C views, gradients SUMMED into ``self.grads`` (layout of grad_step), or
        into a second buffer of the same layout (slot = 1: the data-parallel driver reduces one half step while
        the other is computed)."""
        if self.capacity == 0:
            self.ensure_capacity()
        return self._batched_raw(list(views), list(wmaps), False, slot, journalled)

    # ------------------------------------------------------------------ overflow: journal, snapshot, replay
    _SNAP_TENSORS = ("means", "log_scales", "quats", "logit_opacities", "adam_m", "adam_v", "absgrads", "loss_acc")

    def _snapshot(self) -> None:
        # (round 6: into buffers that are kept between windows, by ONE multi-tensor copy -- eight `clone()`s were eight
        # allocations + eight launches of HOST time in front of the first step of every window, with the GPU idle behind a
        # read-back: ~120 us, 6 us per step of a 20-step window at 100 k Gaussians)
        srcs = [getattr(self, k) for k in self._SNAP_TENSORS]
        bufs = getattr(self, "_snap_bufs", None)
        if bufs is None or any(b.shape != t.shape or b.device != t.device for b, t in zip(bufs, srcs)):
            bufs = self._snap_bufs = [torch.empty_like(t) for t in srcs]
        torch._foreach_copy_(bufs, srcs)
        self._snap = dict(zip(self._SNAP_TENSORS, bufs))
        self._snap["scalars"] = (self.adam_step, list(self.group_steps), self.step, self.absgrads_normalize_factor,
                                 self.epoch, self.loss_scale)

    def _restore(self) -> None:
        self._drop_projection()
        for k in self._SNAP_TENSORS:
            getattr(self, k).copy_(self._snap[k])  # in place: the cached argument block keeps its pointers
        (self.adam_step, gs, self.step, self.absgrads_normalize_factor, self.epoch, self.loss_scale) = self._snap["scalars"]
        self.group_steps = list(gs)

    def _grow_isect(self, factor: float = 2.0) -> None:
        seg = int(self.seg_cap * factor) // 128 * 128 if self.seg_cap else 0
        if self.T * seg > (1 << 28):
            seg = 0  # absurd fixed segments: the count / scan / emit layout takes over
        # (the record table of the XCD-aware placement spans 8 x the longest per-XCD list, which `capacity` does not
        # bound: an overflow raised there is only cured by growing the table itself)
        self._rec_need = int(getattr(self, "_rec_need", 0) * factor)
        if self.segmented and _lib.load().eg_record_xcd_shift(self.T) > 0:
            self._rec_need = max(self._rec_need, int(self.max_items * factor))  # (whatever it was sized from: the table itself grows)
        self._alloc_isect(int(self.capacity * factor), seg)

    def _recover_from_overflow(self) -> None:
        """Called with the stream drained and a sticky device flag raised: some step since the last read-back
        dropped intersections (buffers too small) or hit a transmittance stop while the re-walk launch was being
        skipped.  Grow / switch the re-walk on, put the state back, run the journalled steps again; repeat until
        clean."""
        def flags():
            # (missed, stalled, overflowed) -- of ANY rank under data parallelism: every rank repairs what any rank
            # tripped over, and every rank raises when one must (nobody is left waiting in a collective)
            (missed, stall), over = self._ctl_bits(), self.overflowed()
            if self._dp is not None and self._dp.world > 1:
                (missed, stall, over), _ = self._dp.reduce_words([int(missed), int(stall), int(over)], [])
            return bool(missed), bool(stall), bool(over)

        missed, stall, over = flags()
        if not (self.replay_on_overflow and self._journal and self._snap is not None):
            self.clear_overflow()
            self.tile_counts.zero_()
            for b in self._batches.values():
                b["tile_counts"].zero_()
            if stall and not over:
                for w in self._ctl_words():
                    w.view(-1, 2)[:, 1].zero_()
                raise RuntimeError(self._STALL_MSG + " and cannot be replayed (journal off or data-parallel leg)")
            self._grow_isect(2.0)  # leave usable buffers behind for a caller that catches and restarts
            raise IsectOverflow("tile-intersection buffers overflowed and the steps since the last read-back "
                                "cannot be replayed (journal off or data-parallel leg): results are invalid; "
                                "buffers were grown, restart from the last checkpoint")
        journal = list(self._journal)
        epoch_now, ls_now = self.epoch, self.loss_scale
        assert 2 * len(journal) <= _lib.MAX_WS_TAG, "journal longer than the tag range"
        self._replaying = True
        try:
            for attempt in range(9):
                if attempt > 0:
                    missed, stall, over = flags()
                    if not (missed or stall or over):
                        self.epoch, self.loss_scale = epoch_now, ls_now
                        return
                    if attempt == 8:
                        break
                if stall and not over:  # (a stall next to an overflow is the overflow's: truncated item tables)
                    raise RuntimeError(self._STALL_MSG)
                if missed:
                    self.rewalk_hint = -1  # stops exist: launch the re-walk from now on (the next read-back sizes it)
                    for b in self._batches.values():
                        b["rewalk_hint"] = -1
                    self.rewalk_misses += 1
                if missed or stall:
                    for w in self._ctl_words():
                        w.view(-1, 2)[:, 1].zero_()
                    o = 4 * (self.T + self.max_items + 3)
                    self.workspace[o:o + 4].zero_()
                    for b in self._batches.values():
                        b["workspace"][:, o:o + 4].zero_()
                if over:
                    self.overflow_events += 1
                    self._grow_isect(2.0)  # (drops the batched work buffers as well: re-allocated, flags clear)
                # the replayed steps draw fresh tags: reserve the whole journal's now (two per entry at most), so that the
                # range cannot wrap -- flush, zero the workspaces -- in the middle of the replay
                if self.chained_forward and self._ws_tag + 2 * len(journal) > _lib.MAX_WS_TAG:
                    self._zero_workspaces()
                self.total.zero_()
                self.tile_counts.zero_()  # (a step that ran out of items leaves the cursors of the unserved tiles behind)
                for b in self._batches.values():
                    b["total"].zero_()
                    b["tile_counts"].zero_()
                self._restore()  # (the running loss sum included)
                for kind, view, wmap, epoch, ls in journal:
                    self.epoch, self.loss_scale = epoch, ls
                    if kind == "1":
                        self._step_raw(view, wmap)
                    elif kind == "r":
                        self._regulariser_raw(view, self.loss_acc[0], *wmap)
                    elif kind == "e":
                        self._mark_raw(view)
                    elif kind == "d":  # a data-parallel step: every rank replays it, the collective included
                        self._dp._step_raw(view, wmap[0], wmap[1])
                    else:
                        self._batched_raw(view, wmap, True)
        finally:
            self._replaying = False
        raise IsectOverflow("tile-intersection buffers still overflow after 8 doublings")

    def _journal_push(self, entry, reserve: bool = True) -> None:
        """(kind, a, b, epoch, loss_scale): snapshot the state in front of the first journalled step of a window."""
        if reserve:
            self._reserve_tags(2)  # (a data-parallel step with two half batches takes two)
        if self.replay_on_overflow:
            if not self._journal:
                self._snapshot()
            self._journal.append(entry)

    def journal_bytes(self) -> int:
        """Bytes of the distinct weight maps the journal keeps alive (the per-step `bg_edge_ratio` draws are fresh
        tensors: at 1600 x 1200 a window of 8 epochs holds ~0.7 GB of them); `train()` reads back early when this
        grows past 512 MB."""
        seen, total = set(), 0
        def tensors(x):
            if isinstance(x, Tensor):
                yield x
            elif isinstance(x, (list, tuple)):
                for y in x:
                    yield from tensors(y)

        for entry in self._journal:
            for t in tensors(entry[2]):
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    total += t.numel() * t.element_size()
        return total

    def _scene_stream(self):
        """Context of the stream this trainer's steps were last enqueued on by `train_steps_multi` (a no-op context for a
        trainer that only ever ran on the current stream): read-backs and replays must follow the scene's own work."""
        import contextlib
        st = getattr(self, "_bound_stream", None)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def flush(self) -> None:
        """Drain the stream, verify that no step since the last read-back overflowed (repairing it if one
        did) and forget the journal.  Every operation that changes the state outside train_step calls it.
        (A trainer driven by `train_steps_multi` does this on the scene's stream.)"""
        with self._scene_stream():
            r = self._read_words()
            if r["overflow"] or r["missed"]:
                self._recover_from_overflow()
            self._journal.clear()

    # ------------------------------------------------------------------ epoch marks (no host sync)
    def _mark_raw(self, k: int) -> None:
        self._loss_buf[1 + k:2 + k].copy_(self.loss_acc)
        self.loss_acc.zero_()

    def mark_epoch(self) -> int:
        """Device-side end of an epoch: park the running projection-loss sum in the next slot and zero the
        accumulator -- no read-back.  `pop_losses()` later returns the parked sums in order.  The mark is journalled
        like a step, so an overflow replay across several epochs reproduces every sum."""
        k = self._n_marks
        if 1 + k >= self._loss_buf.numel():
            raise RuntimeError("mark_epoch: 64 epochs are parked; call pop_losses()")
        self._mark_raw(k)
        if self.replay_on_overflow and self._journal:
            self._journal.append(("e", k, None, self.epoch, self.loss_scale))
        self._n_marks = k + 1
        return k

    def train_step_staged(self, view: int, wmap: Tensor, mark=None) -> None:
        """The same step as ``train_step`` but sequenced from Python, one C-ABI call per stage, with
        ``mark(stage_name)`` called after each enqueue -- bench.py brackets the stages with HIP
        events through it.  Results are identical to ``train_step`` (same kernels, same order)."""
        if self.capacity == 0:
            self.ensure_capacity()
        mark = mark or (lambda name: None)
        self._advance_all()
        self._set_hyper()
        fl = (_lib.FLAG_LOG_SCALES | _lib.FLAG_LOGIT_OPACITIES | _lib.FLAG_ANTIALIASED |
              _lib.FLAG_TIGHT_TILES)
        st = stream()
        vm, K = ptr(self.viewmats[view]), ptr(self.Ks[view])
        N, W, H = self.N, self.width, self.height
        mark("start")
        call("eg_project_fwd", ptr(self.means), ptr(self.quats), ptr(self.log_scales), ptr(self.logit_opacities),
             vm, K, N, W, H, 0.01, 1e10, 0.3, 0.0, fl, ptr(self.splat), None, None, None, None, None, None,
             ptr(self.tile_counts), None, st)
        mark("project_fwd")
        call("eg_tile_offsets", ptr(self.tile_counts), self.T, self.capacity, ptr(self.offsets),
             ptr(self.item_offsets), ptr(self.total), st)
        mark("tile_offsets")
        call("eg_tile_emit", None, None, None, ptr(self.splat), fl, N, W, H, ptr(self.offsets), ptr(self.tile_counts),
             self.capacity, ptr(self.keys), None, st)
        mark("tile_emit")
        call("eg_sort_pairs", ptr(self.keys), ptr(self.offsets), self.T, self.capacity, ptr(self.flatten_ids),
             None, getattr(self, "max_tile_seen", 0), st)
        mark("tile_sort")
        call("eg_composite_fwd", ptr(self.splat), None, 1, ptr(self.offsets), ptr(self.flatten_ids), W, H,
             ptr(self.render), ptr(self.alphas), ptr(self.last_ids), ptr(self.gt[view]), ptr(wmap),
             self.loss_scale, ptr(self.vpix), ptr(self.loss_acc), ptr(self.item_offsets), ptr(self.total),
             self.max_items, ptr(self.workspace), ptr(self.gtstop), self.rewalk_hint, st)
        mark("composite_fwd")
        call("eg_backward_fused", ptr(self.means), ptr(self.quats), ptr(self.log_scales),
             ptr(self.logit_opacities), vm, K, N, W, H, 0.3, fl, ptr(self.splat), ptr(self.gtstop), ptr(self.g2d),
             None, None, None, None, ptr(self.absgrads), ptr(self.adam_m), ptr(self.adam_v), C.byref(self._hyper), st)
        mark("backward_fused")
        self.absgrads_normalize_factor += 1
        self.step += 1

    @staticmethod
    def timing_begin(n_steps: int) -> None:
        """The next `n_steps` train_step / grad_step calls record HIP events between their stages
        (natively, on the launch stream)."""
        rc = _lib.load().eg_timing_begin(n_steps)
        if rc != 0:
            raise RuntimeError(_lib.load().eg_last_error_string().decode())

    @staticmethod
    def timing_end() -> Dict[str, float]:
        """Synchronises once; average launch duration in microseconds per stage."""
        lib = _lib.load()
        k = lib.eg_timing_stage_count()
        buf = (C.c_float * k)()
        n = C.c_int32(0)
        rc = lib.eg_timing_end(buf, C.byref(n))
        if rc != 0:
            raise RuntimeError(lib.eg_last_error_string().decode())
        return {lib.eg_timing_stage_name(i).decode(): float(buf[i]) for i in range(k)}

    def grad_step(self, view: int, wmap: Tensor, journalled: bool = False) -> Tensor:
        """Forward + loss + backward only: leaves dL/d{means,quats,log_scales,logit_opacities} in
        ``self.grads`` ([means 3N | quats 4N | scales 3N | opac N | absgrad increment N] flat
        blocks) for the data-parallel driver, which all-reduces them and then calls ``apply_adam``.
        journalled: the caller keeps a journal of its steps (DataParallelStep does), so the forward may speculate that no
        pixel reaches the transmittance stop, like train_step."""
        if self.capacity == 0:
            self.ensure_capacity()
        a = self._args(view, wmap, False)
        if journalled:
            a.rewalk_hint = self._rewalk_arg(True)
        # apply_adam(next_view=view) has already projected + binned this view with the parameters it produced
        a.have_projection = 1 if (self._projected == view and self.seg_cap) else 0
        if a.have_projection:
            self._projected = None  # consumed: the step's sort returns the cursors to zero
        else:
            self._drop_projection()
        call("eg_train_step", C.byref(a), stream())
        a.have_projection = 0
        self.step += 1
        return self.grads

    def _dp_steps_raw(self, views: List[int], wmaps: List[Tensor], next_view: Optional[int], journalled: bool) -> None:
        """K data-parallel steps of this rank by one native call (eg_train_steps_dp: grad -> RCCL all-reduce on the launch
        stream -> Adam + projection of the next view); the host-side bookkeeping of K x (grad_step, apply_adam)."""
        K = len(views)
        if self.capacity == 0:
            self.ensure_capacity()
        for w in wmaps:
            assert w.is_cuda and w.is_contiguous() and w.shape == (self.height, self.width)
        self._advance_all()   # step 0's counts; the native loop advances them by k
        self._set_hyper()
        a = self._args(views[0], wmaps[0], False, n_tags=K)
        if journalled:
            a.rewalk_hint = self._rewalk_arg(True)
        a.have_projection = 1 if (self._projected == views[0] and self.seg_cap) else 0
        if a.have_projection:
            self._projected = None
        else:
            self._drop_projection()
        va = (C.c_int32 * K)(*views)
        wa = (C.c_void_p * K)(*[w.data_ptr() for w in wmaps])
        call("eg_train_steps_dp", C.byref(a), C.byref(self._hyper), ptr(self.absgrads), K, va, wa, ptr(self.viewmats),
             ptr(self.Ks), ptr(self.gt), -1 if next_view is None else int(next_view), stream())
        a.have_projection = 0
        for _ in range(K - 1):
            self._advance_all()
        self.absgrads_normalize_factor += K
        self.step += K
        self._projected = None if next_view is None else int(next_view)

    def grad_views(self):
        N, g = self.N, self.grads.view(-1)
        return (g[:3 * N].view(N, 3), g[3 * N:7 * N].view(N, 4), g[7 * N:10 * N].view(N, 3), g[10 * N:11 * N])

    def apply_adam(self, next_view: Optional[int] = None) -> None:
        """The four Adam steps on the (all-reduced) gradient buffer.  next_view: the view this rank rasterises next
        -- Adam and that view's projection + binning then run as ONE launch (eg_adam_emit) and the following
        `grad_step(next_view)` skips its projection."""
        self._advance_all()
        self._set_hyper()
        c = self._args_cache.get("adam_ptrs")
        if c is None:  # raw pointers of the gradient blocks (rebuilt with the buffers): no tensor views per step
            g0, N = self.grads.data_ptr(), self.N
            c = (ptr(self.means), ptr(self.log_scales), ptr(self.quats), ptr(self.logit_opacities),
                 g0, g0 + 4 * 7 * N, g0 + 4 * 3 * N, g0 + 4 * 10 * N, ptr(self.adam_m), ptr(self.adam_v), g0 + 4 * 11 * N)
            self._args_cache["adam_ptrs"] = c
        self._drop_projection()
        if next_view is not None and self.seg_cap and self.capacity:
            fl = (_lib.FLAG_LOG_SCALES | _lib.FLAG_LOGIT_OPACITIES | _lib.FLAG_ANTIALIASED | _lib.FLAG_TIGHT_TILES)
            if self.T > _lib.PREFIX_HERE_MAX_TILES:  # (what eg_train_step's own projection does on such a grid)
                fl |= _lib.FLAG_FRONT_PREFIX
            call("eg_adam_emit", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], self.N, self._hyper,
                 c[10], ptr(self.absgrads), self.viewmats.data_ptr() + 64 * next_view, self.Ks.data_ptr() + 36 * next_view,
                 self.width, self.height, fl, ptr(self.splat), ptr(self.tile_counts), self.seg_cap, ptr(self.keys),
                 ptr(self.item_offsets), self.max_items, ptr(self.total),
                 ptr(self.ticket) if self.T > _lib.PREFIX_HERE_MAX_TILES else None, stream())  # (small grids: the
            # following step's sort kernel forms the tile prefix itself: no scan tail here)
            self._projected = int(next_view)
        else:
            call("eg_adam_multi", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], self.N, self._hyper,
                 c[10], ptr(self.absgrads), stream())  # += all-reduced absgrad block
        self.absgrads_normalize_factor += 1

    # ------------------------------------------------------------------ orientation regularisers (8f)
    def update_nearest_neighbors(self, dir_loss_num_nn: int = 5, enforce_method: str = "enforce_full") -> Tensor:
        """`update_nearest_neighbors` (edge_gs.py:326-344) on device; keeps the reference's quirk of
        skipping the nearest neighbour (see regularizers.reference_nn_indices).  No host sync: the search grid
        (bounding box of the means) is chosen on the device."""
        from . import regularizers as R
        n = 2 * dir_loss_num_nn + 1 if enforce_method == "enforce_half" else dir_loss_num_nn + 1
        buf = self.__dict__.get("_knn_buf")
        if buf is None or buf.shape != (self.N, n):  # the table is reused from call to call
            buf = self._knn_buf = torch.empty(self.N, n, dtype=torch.int32, device=self.dev)
            # every Gaussian's K-th squared distance of the previous search: the means move a little between two
            # regulariser steps, so it is a tight entry bound for the next one (zeros = unknown; dropped whenever the
            # rows change: N, spatial re-sort)
            self._knn_kth = torch.zeros(self.N, device=self.dev)
        # exhaustive (small N) or device-chosen grid: no host sync either way
        R.knn(self.means, n, out=buf, kth=self._knn_kth if self.N > R.KNN_EXHAUSTIVE_MAX else None)
        self.nn_table = buf              # [N, n]: k_nearest_sklearn's table (self excluded)
        self.nn_indices = buf[:, 1:]     # the reference then drops the nearest neighbour too (edge_gs.py:342)
        return self.nn_indices

    def regulariser_step(self, kind: str, avg_loss_sum=None, scale_factor: float = 0.01,
                         dir_loss_num_nn: int = 5, enforce_method: str = "enforce_full", want_value: bool = True):
        """One regulariser iteration of train_gaussians.py:108-131 ('direction' or 'ratio'):
        loss -> lambda = avg_loss_sum * scale_factor / loss -> backward -> Adam step of the
        means / scales / quats optimizers only (their step counts advance, the opacity optimizer's does
        not).  With the reference's torch 1.13 `zero_grad()` (zeroed, not None) the optimizers whose
        parameter is outside the loss still step on a zero gradient; that is reproduced.

        avg_loss_sum = None: the running sum of this epoch's projection losses is taken from the DEVICE accumulator
        (what `pop_loss` would return) and lambda is formed on the device -- no host sync at all (the reference
        has two `.item()` per regulariser step); the step is journalled for overflow replay like a train_step.
        A float: the caller's host value (one sync for the journal check).  Returns the loss value as a float
        when want_value (one sync), else the device scalar."""
        device_lambda = avg_loss_sum is None
        if device_lambda and self.replay_on_overflow:
            if not self._journal:
                self._snapshot()
            self._journal.append(("r", kind, (scale_factor, dir_loss_num_nn, enforce_method), self.epoch, self.loss_scale))
        elif self._journal:
            self.flush()
        loss = self._regulariser_raw(kind, self.loss_acc[0] if device_lambda else avg_loss_sum, scale_factor,
                                     dir_loss_num_nn, enforce_method)
        return float(loss) if want_value else loss

    def _regulariser_raw(self, kind, avg_loss_sum, scale_factor, dir_loss_num_nn, enforce_method):
        """kNN (direction) + ONE native enqueue (eg_regulariser_step: loss, lambda on the device, backward, Adam).
        avg_loss_sum: a device scalar tensor or a host float.  Returns the loss value as a device scalar."""
        self._drop_projection()
        if kind not in ("direction", "ratio"):
            raise ValueError(f"unknown regulariser: {kind}")
        K = top_k = 0
        nn = None
        if kind == "direction":
            self.update_nearest_neighbors(dir_loss_num_nn, enforce_method)
            nn, K = self.nn_table, self.nn_table.shape[1] - 1
            top_k = dir_loss_num_nn if enforce_method == "enforce_half" else 0
        self.group_steps = [self.group_steps[0] + 1, self.group_steps[1] + 1, self.group_steps[2] + 1,
                            self.group_steps[3]]
        self._set_hyper()
        self._hyper.group_steps[3] = -1  # the opacity optimizer does not step here (train_gaussians.py:116-119)
        work = self.__dict__.get("_reg_work")
        if work is None:
            work = self._reg_work = torch.zeros(2, device=self.dev)
        dev_sum = isinstance(avg_loss_sum, Tensor)
        if dev_sum and self._dp is not None and self._dp.world > 1:
            # data parallel: the epoch's running loss sum is the sum over the ranks' views (every replica forms the
            # same lambda, train_gaussians.py:113,125); a copy is reduced, the local accumulator keeps its own share
            avg_loss_sum = self._dp.all_reduce_(avg_loss_sum.detach().clone().reshape(1))
        # Under data parallelism every rank takes this step from the same state and must END in the same state (densify /
        # cull decisions, the sharded views' gradients): the neighbour gradients and the loss sums are then accumulated in
        # 64-bit fixed point (eg_regulariser_step_fixed: integer atomics are order-independent), so the replicas stay
        # bit-identical without exchanging anything -- round 3 summed with float atomics and had rank 0 broadcast its
        # parameters and moments after every regulariser step (128 bytes per Gaussian).  `deterministic_regularisers`
        # asks for the same on a single GPU (run-to-run reproducibility; one more small launch).
        if (self._dp is not None and self._dp.world > 1) or getattr(self, "deterministic_regularisers", False):
            fx = self.__dict__.get("_reg_fixed")
            if fx is None or fx.numel() != 3 * self.N + 1:
                fx = self._reg_fixed = torch.zeros(3 * self.N + 1, dtype=torch.int64, device=self.dev)
            call("eg_regulariser_step_fixed", 0 if kind == "direction" else 1, ptr(self.means), ptr(self.quats),
                 ptr(self.log_scales), ptr(self.logit_opacities), ptr(self.adam_m), ptr(self.adam_v), ptr(self.grads), self.N,
                 ptr(nn) if nn is not None else None, K + 1, 1, K, top_k, ptr(avg_loss_sum) if dev_sum else None,
                 0.0 if dev_sum else float(avg_loss_sum), float(scale_factor), ptr(work), self._hyper, ptr(fx), stream())
        else:
            call("eg_regulariser_step", 0 if kind == "direction" else 1, ptr(self.means), ptr(self.quats),
                 ptr(self.log_scales), ptr(self.logit_opacities), ptr(self.adam_m), ptr(self.adam_v), ptr(self.grads), self.N,
                 ptr(nn) if nn is not None else None, K + 1, 1, K, top_k, ptr(avg_loss_sum) if dev_sum else None,
                 0.0 if dev_sum else float(avg_loss_sum), float(scale_factor), ptr(work), self._hyper, stream())
        return work[1]

    # ------------------------------------------------------------------ read-backs (these sync)
    def _read_words(self) -> Dict:
        """ONE device->host copy (one sync) of every word the host looks at between runs of steps: the loss sums,
        (M, sticky overflow flag, items, largest tile) and the re-walk control words (longest exact-stop list seen,
        missed-re-walk flag) of the single-view buffers and of every batched set in use."""
        o = 4 * (self.T + self.max_items + 2)
        n = 1 + self._n_marks
        parts = [self._loss_buf[:n].view(torch.int32), self.total, self.workspace[o:o + 8].view(torch.int32)]
        for b in self._batches.values():
            parts.append(b["total"].max(dim=0).values)
            parts.append(b["workspace"][:, o:o + 8].contiguous().view(torch.int32).view(-1, 2).max(dim=0).values)
        host = torch.cat(parts).cpu()
        sums = host[:n].view(torch.float32).tolist()
        words = host[n:].tolist()
        tot, seen, missed = words[0:4], words[4], words[5]
        batch_seen = []
        for i in range(len(self._batches)):
            w = words[6 + 6 * i:12 + 6 * i]
            tot = [max(a, x) for a, x in zip(tot, w[:4])]
            batch_seen.append(w[4])
            missed = max(missed, w[5])
        if self._dp is not None and self._dp.world > 1:
            # every rank must take the same decisions (replay or not, buffer sizes, launch modes): flags and shape hints
            # by max, the loss sums by sum -- two small collectives per read-back
            ints = [tot[0], tot[1], tot[3], seen, missed] + list(batch_seen)
            ints, sums = self._dp.reduce_words(ints, sums)
            tot = [ints[0], ints[1], tot[2], ints[2]]
            seen, missed, batch_seen = ints[3], ints[4], list(ints[5:])
        return {"acc": sums[0], "marks": sums[1:], "m_last": tot[0], "overflow": tot[1] != 0, "tile_max": tot[3],
                "seen": seen, "batch_seen": batch_seen, "missed": missed != 0}

    def _sync_state(self) -> Dict:
        with self._scene_stream():  # (train_steps_multi: on the scene's stream)
            return self._sync_state_here()

    def _sync_state_here(self) -> Dict:
        """The read-back between runs of steps: checks the sticky flags (replaying the journalled steps when one is
        raised), forgets the journal, zeroes the loss sums and refreshes launch-shape hints and buffer sizes."""
        r = self._read_words()
        if r["overflow"] or r["missed"]:  # sticky flags: SOME step since the last read-back must be repeated
            self._recover_from_overflow()  # raises IsectOverflow when the steps cannot be replayed
            r = self._read_words()
        self._journal.clear()
        self.loss_acc.zero_()
        self._n_marks = 0
        # the stream is drained anyway: refresh the launch-shape hints -- the longest exact-stop re-walk list since
        # the last read-back (control word 2 of the compositing workspace) ...
        o = 4 * (self.T + self.max_items + 2)
        if r["seen"]:
            self.workspace[o:o + 4].zero_()
        # (hysteresis: a window without a stop does not send the forward back to its speculative mode at once -- a scene
        # whose pixels stop now and then would pay a replayed window at every relapse; four calm windows do)
        self._calm_windows = 0 if r["seen"] else getattr(self, "_calm_windows", 4) + 1
        seen = r["seen"] if (r["seen"] or self._calm_windows >= 4 or self.rewalk_hint <= 0) else self.rewalk_hint
        if seen != self.rewalk_hint:
            self.rewalk_hint = seen
            self._args_cache = {}
        for b, seen in zip(self._batches.values(), r["batch_seen"]):
            if seen or self._calm_windows >= 4 or b["rewalk_hint"] <= 0:
                b["rewalk_hint"] = seen
            if seen:
                b["workspace"][:, o:o + 4].zero_()
        # ... and the tile-sort launch hint from the last step's scan
        m_last, tile_max = r["m_last"], r["tile_max"]
        if tile_max > getattr(self, "max_tile_seen", 0):
            self.max_tile_seen = tile_max
            self._args_cache = {}
        # ... and grow ahead of the drift (opacities climbing, Gaussians converging on the edges)
        seg = self.seg_cap
        if seg and tile_max * 1.15 > seg:  # a tile is about to outgrow its segment
            seg = (int(tile_max * 1.5) // 128 + 2) * 128
            if self.T * seg > (1 << 28):
                seg = 0
        cap = self.capacity
        if m_last * 1.15 > cap:
            cap = int(m_last * 1.5) + 4096
        if seg != self.seg_cap or cap != self.capacity:
            self._alloc_isect(cap, seg)
        return r

    def pop_loss(self) -> float:
        """Sum of the projection losses since the last read-back (the reference's avg_loss numerator,
        train_gaussians.py:99) -- ONE device sync for many steps instead of two per step."""
        r = self._sync_state()
        return float(sum(r["marks"]) + r["acc"])

    def pop_losses(self):
        """(sums of the epochs parked by mark_epoch since the last read-back, in order; the sum of the steps after the
        last mark) -- one device sync for many EPOCHS."""
        r = self._sync_state()
        return [float(x) for x in r["marks"]], float(r["acc"])

    def _totals(self):
        """(M, sticky overflow flag, items, largest tile) of the last step(s): single-view buffers and, when a
        batched step ran, the maximum over its views."""
        t = [int(x) for x in self.total.tolist()]
        for b in self._batches.values():
            bt = b["total"].max(dim=0).values.tolist()
            t = [max(a, int(x)) for a, x in zip(t, bt)]
        return t

    def overflowed(self) -> bool:
        """True if ANY step since the flag was last cleared dropped intersections (the device flag is sticky)."""
        return self._totals()[1] != 0

    def clear_overflow(self) -> None:
        self.total[1:2].zero_()
        for b in self._batches.values():
            b["total"][:, 1].zero_()

    def fused_backward_active(self) -> bool:
        """Inside a native run of steps (`train_steps`) the backward of a step is ONE kernel (csrc/backward_fused.hip)."""
        return bool(self.segmented and self.seg_cap > 0 and not self.two_kernel_backward
                    and _lib.load().eg_backward_is_fused(self.N, self.T))

    def last_m(self) -> int:
        return self._totals()[0]

    # ------------------------------------------------------------------ densify / cull
    def _moment_views(self, t: Tensor):
        N = self.N
        return {"means": t[:3 * N].view(N, 3), "scales": t[3 * N:6 * N].view(N, 3),
                "quats": t[6 * N:10 * N].view(N, 4), "opacities": t[10 * N:11 * N].view(N, 1)}

    def _params(self):
        return {"means": self.means, "scales": self.log_scales, "quats": self.quats,
                "opacities": self.logit_opacities.view(-1, 1)}

    def _scan(self, mask_u8: Tensor):
        pos = torch.empty(self.N, dtype=torch.int32, device=self.dev)
        cnt = torch.empty(1, dtype=torch.int32, device=self.dev)
        call("eg_mask_scan", ptr(mask_u8), self.N, ptr(pos), ptr(cnt), stream())
        return pos, int(cnt.item())

    # ------------------------------------------------------------------ row order (data layout)
    def _morton_perm(self) -> Tensor:
        m = self.means
        lo, hi = m.min(0).values, m.max(0).values
        q = ((m - lo) / (hi - lo + 1e-12) * 1023.0).long().clamp_(0, 1023)

        def spread(v):  # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        return torch.argsort(code, stable=True)

    def spatial_sort(self) -> None:
        """Permute every per-Gaussian array (parameters, Adam moments, absgrads) into Morton order of
        the current means.  A pure relabelling: the step treats Gaussians independently."""
        self._drop_projection()
        N = self.N
        if N == 0:
            return
        if self._journal:
            self.flush()
        perm = self._morton_perm()
        m_old, v_old = self._moment_views(self.adam_m), self._moment_views(self.adam_v)
        names = list(self._params().keys())
        self.adam_m = torch.cat([m_old[k][perm].reshape(-1) for k in names]).contiguous()
        self.adam_v = torch.cat([v_old[k][perm].reshape(-1) for k in names]).contiguous()
        self.means = self.means[perm].contiguous()
        self.log_scales = self.log_scales[perm].contiguous()
        self.quats = self.quats[perm].contiguous()
        self.logit_opacities = self.logit_opacities[perm].contiguous()
        self.absgrads = self.absgrads[perm].contiguous()
        ref = self.ref_index if self.ref_index is not None else torch.arange(N, device=self.dev)
        self.ref_index = ref[perm].contiguous()
        self.nn_indices = None
        self._alloc_per_gaussian()

    def _in_reference_order(self, t: Tensor) -> Tensor:
        """Rows of a per-Gaussian tensor put back where the reference's arrays hold them."""
        if self.ref_index is None:
            return t
        out = torch.empty_like(t)
        out[self.ref_index] = t
        return out

    def cull(self, cull_mask: Tensor, reset_opacity_value: float = 0.08) -> int:
        """cull_gaussians (edge_gs.py:412-429) + remove_from_all_optim (:384-409): rows of the 4
        params, 8 moment tensors and absgrads where ~cull_mask, then the reference's opacity clamp
        (a probability-space constant applied in LOGIT space -- kept, it is what the reference does)."""
        if self._journal:
            self.flush()
        keep = (~cull_mask.to(self.dev).bool()).to(torch.uint8).contiguous()
        pos, n_keep = self._scan(keep)
        N = self.N
        m_old, v_old = self._moment_views(self.adam_m), self._moment_views(self.adam_v)
        new_m = torch.zeros(11 * n_keep, device=self.dev)
        new_v = torch.zeros(11 * n_keep, device=self.dev)
        new_p = {}
        off = 0
        for name, p in self._params().items():
            d = p.shape[1]
            out = torch.empty(n_keep, d, device=self.dev)
            call("eg_compact_rows", ptr(p), ptr(keep), ptr(pos), N, d, ptr(out), stream())
            new_p[name] = out
            for old, new in ((m_old[name], new_m), (v_old[name], new_v)):
                dst = new[off:off + n_keep * d].view(n_keep, d)
                call("eg_compact_rows", ptr(old.contiguous()), ptr(keep), ptr(pos), N, d, ptr(dst), stream())
            off += n_keep * d
        ag = torch.empty(n_keep, 1, device=self.dev)
        call("eg_compact_rows", ptr(self.absgrads.view(-1, 1)), ptr(keep), ptr(pos), N, 1, ptr(ag), stream())
        self.means, self.log_scales, self.quats = new_p["means"], new_p["scales"], new_p["quats"]
        self.logit_opacities = new_p["opacities"].view(-1).clamp_(max=reset_opacity_value)
        self.adam_m, self.adam_v, self.absgrads = new_m, new_v, ag.view(-1)
        if self.ref_index is not None:  # the reference compacts its own arrays: ranks among the survivors
            kept = self.ref_index[keep.bool()]
            ranks = torch.empty_like(kept)
            ranks[torch.argsort(kept)] = torch.arange(kept.numel(), device=self.dev)
            self.ref_index = ranks
        self._alloc_per_gaussian()
        return N - n_keep

    def cull_opacity(self, value: float = 0.05, reset_opacity_value: float = 0.08) -> int:
        """cull_gaussians_opacity, 'absolute' (edge_gs.py:477-488; every shipped config)."""
        if self._journal:
            self.flush()  # the mask must come from the verified state (steps may still be rolled back and replayed)
        return self.cull(torch.sigmoid(self.logit_opacities) < value, reset_opacity_value)

    def duplicate(self, dup_mask: Tensor, dup_factor: int = 3, noise_scale: float = 0.05,
                  noise: Optional[Tensor] = None) -> int:
        """dup_gaussians (edge_gs.py:460-474) + dup_in_all_optim (:431-457): append dup_factor-1
        copies of the masked rows; means get N(0, noise_scale^2) noise, moments of new rows are 0."""
        if self._journal:
            self.flush()
        sel = dup_mask.to(self.dev).bool().to(torch.uint8).contiguous()
        pos, n_sel = self._scan(sel)
        copies = dup_factor - 1
        N, n_new = self.N, self.N + copies * n_sel
        if noise is None:
            gen = torch.Generator(device=self.dev).manual_seed(self.seed * 1_000_003 + self._dup_events)
            noise = torch.randn(copies * n_sel, 3, device=self.dev, generator=gen)
        self._dup_events += 1
        new_ref = None
        if self.ref_index is not None and n_sel:
            # the reference appends copy k of its j-th selected row at N + k n_sel + j: our j-th selected
            # row is its rank-th one; a caller-supplied `noise` is in the reference's order
            sel_ref = self.ref_index[sel.bool()]
            rank = torch.empty_like(sel_ref)
            rank[torch.argsort(sel_ref)] = torch.arange(n_sel, device=self.dev)
            new_ref = torch.cat([N + k * n_sel + rank for k in range(copies)])
            noise = noise.to(self.dev)[new_ref - N]
        noise = (noise.to(self.dev) * noise_scale).contiguous()
        m_old, v_old = self._moment_views(self.adam_m), self._moment_views(self.adam_v)
        new_m = torch.zeros(11 * n_new, device=self.dev)
        new_v = torch.zeros(11 * n_new, device=self.dev)
        new_p = {}
        off = 0
        for name, p in self._params().items():
            d = p.shape[1]
            out = torch.empty(n_new, d, device=self.dev)
            out[:N] = p
            if n_sel:
                call("eg_append_rows", ptr(p.contiguous()), ptr(sel), ptr(pos), N, n_sel, d, copies,
                     ptr(noise) if name == "means" else None, 0.0, ptr(out[N:]), stream())
            new_p[name] = out
            new_m[off:off + N * d] = m_old[name].reshape(-1)
            new_v[off:off + N * d] = v_old[name].reshape(-1)
            off += n_new * d
        self.means, self.log_scales, self.quats = new_p["means"], new_p["scales"], new_p["quats"]
        self.logit_opacities = new_p["opacities"].view(-1)
        self.adam_m, self.adam_v = new_m, new_v
        if new_ref is not None:
            self.ref_index = torch.cat([self.ref_index, new_ref])
        self.reset_absgrads()
        self._alloc_per_gaussian()
        return n_sel

    def duplicate_high_pos_gradients(self, threshold: float = 0.5, dup_factor: int = 3,
                                     noise_scale: float = 0.05, noise: Optional[Tensor] = None,
                                     threshold_type: str = "absolute") -> int:
        """edge_gs.py:544-576.  'absolute': min-max normalised mean absgrad > threshold (every shipped
        config).  'percentile_top' (:559-568): the threshold is the int(1/value)-quantile boundary of the
        RAW mean absgrads ('lower' interpolation) -- and is then compared with the NORMALISED values, as the
        reference does."""
        if self._journal:
            self.flush()  # absgrads of steps that may still be replayed must not decide the mask
        g = self.absgrads / self.absgrads_normalize_factor
        gn = (g - g.min()) / (g.max() - g.min())
        if threshold_type == "absolute":
            mask = gn > threshold
        elif threshold_type == "percentile_top":
            nq = int(1 / threshold)
            thr = torch.quantile(g, (nq - 1) / nq, interpolation="lower") if nq > 1 else torch.zeros((), device=g.device)
            mask = gn > thr
        else:
            raise NotImplementedError(f"dup_threshold_type={threshold_type!r} (edge_gs.py:559-573 defines "
                                      "'absolute' and 'percentile_top')")
        return self.duplicate(mask, dup_factor, noise_scale, noise)

    def reset_absgrads(self):
        self.absgrads = torch.zeros(self.means.shape[0], device=self.dev)
        self.absgrads_normalize_factor = 1

    def cull_not_projecting(self, edge_masks_u8: Tensor, min_projecting_fraction: float = 0.1,
                            reset_opacity_value: float = 0.08) -> int:
        """cull_gaussians_not_projecting (edge_gs.py:578-601) as one N x V device kernel.
        edge_masks_u8: [V,H,W] uint8 (gt >= 0.5)."""
        if self._journal:
            self.flush()
        P = torch.bmm(self.Ks, self.viewmats[:, :3, :4]).contiguous()  # K @ viewmat[:3,:4]
        hits = torch.zeros(self.N, dtype=torch.int32, device=self.dev)
        call("eg_project_hits", ptr(self.means), self.N, ptr(P), self.V, ptr(edge_masks_u8.contiguous()),
             self.width, self.height, ptr(hits), stream())
        frac = hits.float() / float(self.V)
        return self.cull(frac < min_projecting_fraction, reset_opacity_value)

    # ------------------------------------------------------------------ hand-off
    def export_as_ply(self, ply_path: str) -> None:
        """`EdgeGaussianSplatting.export_as_ply` (edge_gs.py:635-642): same file, no plyfile needed."""
        from . import io as egio
        egio.export_as_ply(self.state_dict(), ply_path)

    def load_state_dict(self, state: Dict[str, Tensor]) -> None:
        """Weights only, like the reference's `--ckpt_path` (edge_gs.py:625-633, train_gaussians.py builds
        fresh optimizers after loading): Adam moments, absgrads AND every step counter (Adam bias
        correction, model.step alternation phase, absgrad normaliser) restart."""
        if self._journal:
            self.flush()
        f = dict(device=self.dev, dtype=torch.float32)
        self.means = state["gauss_params.means"].detach().to(**f).contiguous().clone()
        self.log_scales = state["gauss_params.scales"].detach().to(**f).contiguous().clone()
        self.quats = state["gauss_params.quats"].detach().to(**f).contiguous().clone()
        self.logit_opacities = state["gauss_params.opacities"].detach().to(**f).reshape(-1).contiguous().clone()
        self._alloc_state()
        self.adam_step, self.group_steps, self.step = 0, [0, 0, 0, 0], 0
        self.capacity = 0
        self.ref_index = None
        if self.spatial_order:
            self.spatial_sort()

    def state_dict(self) -> Dict[str, Tensor]:
        """Same keys / shapes -- and, whatever the internal row order, the same row order -- as the
        reference's checkpoint (edge_gs.py:625-633)."""
        if self._journal:
            self.flush()  # (speculated / overflowed steps are repaired before anything is exported)
        r = self._in_reference_order
        return {"gauss_params.means": r(self.means).clone(), "gauss_params.scales": r(self.log_scales).clone(),
                "gauss_params.quats": r(self.quats).clone(),
                "gauss_params.opacities": r(self.logit_opacities).view(-1, 1).clone()}


def train_steps_multi(trainers: List["EdgeTrainer"], views: List[List[int]], wmaps: List[List[Tensor]], streams: List,
                      n_threads: int = 0) -> None:
    """K consecutive reference iterations of EACH of S independent scenes, enqueued by ONE native call
    (`eg_train_steps_multi`): scene s = `trainers[s].train_steps(views[s], wmaps[s])` on `streams[s]` (torch.cuda.Stream
    objects, all different, none of them the stream a trainer's tensors are still being written on).  The scenes share
    nothing -- every trainer ends exactly where its solo run ends (tests/test_gpu_parity.py) -- but one GPU runs their
    launch sequences side by side: BASELINE configs[4] ("115-scan sweep, one scene per GPU") with S scenes per device.
    n_threads: host threads inside the native call (0: one per scene, at most 8).  The weight maps must have been produced on
    the scene's stream (or before a synchronisation).  The trainer remembers `streams[s]`: its later read-backs (`flush`,
    `pop_loss`, `pop_losses`) and an overflow replay of these steps run on that stream whatever stream is current; any OTHER
    call on the trainer (densify, single steps) must be made under `torch.cuda.stream(streams[s])` or after a synchronisation.  After a failure of the native call the trainers' states are undefined
    (some scenes have enqueued more steps than others): restore them from checkpoints."""
    S = len(trainers)
    assert S >= 1 and len(views) == S and len(wmaps) == S and len(streams) == S
    K = len(views[0])
    assert all(len(v) == K for v in views) and all(len(w) == K for w in wmaps), "the same number of steps for every scene"
    assert len({int(st.cuda_stream) for st in streams}) == S and len({id(t) for t in trainers}) == S
    if K == 0:
        return
    for tr, vs, ws in zip(trainers, views, wmaps):  # (checked BEFORE any trainer's host state moves)
        for w in ws:
            assert w.is_cuda and w.is_contiguous() and w.shape == (tr.height, tr.width), "weight maps: contiguous [H, W] device tensors"
    blocks = []
    # (round 6: several scenes side by side keep the two-kernel backward -- the one-kernel form is built for four waves per
    # SIMD, which is what a lone 30 k-Gaussian scene wants and what eight scenes sharing the chip do not: 1.49 against 1.32 G
    # Gaussians*views/s at S = 8, profiles/r06_scenes_per_gpu.txt; same parameters either way)
    for tr in trainers:
        tr._side_by_side = S >= 2
    for tr, vs, ws, sx in zip(trainers, views, wmaps, streams):
        with torch.cuda.stream(sx):  # (whatever host-side preparation enqueues -- a tag wrap's zeroing, a snapshot -- goes to the scene's stream)
            if tr.capacity == 0:
                tr.ensure_capacity()
            tr._reserve_tags(K)
            if tr.replay_on_overflow:
                if not tr._journal:
                    tr._snapshot()
                tr._journal.extend(("1", v, w, tr.epoch, tr.loss_scale) for v, w in zip(vs, ws))
            blocks.append(tr._steps_begin(vs, ws))
        tr._bound_stream = sx  # (flush / pop_loss / a replay of these steps run on the scene's stream from now on)
    args = (C.POINTER(_lib.StepArgs) * S)(*[C.pointer(b[0]) for b in blocks])
    va = (C.POINTER(C.c_int32) * S)(*[C.cast(b[1], C.POINTER(C.c_int32)) for b in blocks])
    wa = (C.POINTER(C.c_void_p) * S)(*[C.cast(b[2], C.POINTER(C.c_void_p)) for b in blocks])
    vm = (C.c_void_p * S)(*[ptr(t.viewmats) for t in trainers])
    ks = (C.c_void_p * S)(*[ptr(t.Ks) for t in trainers])
    gt = (C.c_void_p * S)(*[ptr(t.gt) for t in trainers])
    st = (C.c_void_p * S)(*[int(x.cuda_stream) for x in streams])
    try:
        call("eg_train_steps_multi", S, args, K, va, wa, vm, ks, gt, st, n_threads if n_threads > 0 else min(S, 8))
    finally:
        for tr in trainers:
            tr._side_by_side = False
    for tr in trainers:
        tr._steps_end(K)
