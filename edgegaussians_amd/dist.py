"""View-sharded data parallelism: one process per GPU, RCCL all-reduce of Gaussian gradients.

The reference is single-process / single-GPU (no torch.distributed call site) and
steps the optimizer after every single view (train_gaussians.py:104-106,311).  The path shards
naturally over VIEWS: every rank holds the full Gaussian state (11 floats/Gaussian +
Adam moments), takes different views of the step's view batch, and the only exchange is an
all-reduce(sum) over a fused [N,12] fp32 buffer
    [ dL/dmeans 3 | dL/dquats 4 | dL/dlog_scales 3 | dL/dlogit_opacity 1 | absgrad increment 1 ]
after which every rank applies the identical fused Adam -- replicas stay bit-identical because the
all-reduce result is identical on every rank.  This is a THROUGHPUT mode: a P-view batch per
optimizer step is a different trajectory from P sequential steps; what is guaranteed (and tested)
is  all-reduced gradient == sum of the single-GPU per-view gradients at the same parameters.

Backend "nccl" IS RCCL on ROCm; on the MI355X xGMI mesh a 1.4-24 MB buffer (N = 30k-500k) is
latency- to per-link-bound, so it is sent as one collective per half step rather than per-parameter
buckets.  Hiding it: with C >= 2 views per rank and step (`views_per_rank`, the rank's views run as ONE
batched launch sequence, EdgeTrainer.grad_step_batched) the step is split into two half batches whose
gradients land in two buffers: the all-reduce of the first half runs on RCCL's stream while the second
half is being rasterised; only the second, shorter-lived collective is exposed.  With one view per rank
there is nothing to overlap with (the next forward needs the updated parameters) and the collective is
exposed in full -- `comm_us()` reports the exposed time either way.
The same code runs under "gloo" on CPU tensors (tests/test_dist_gloo.py) and, staged through the host,
with several ranks sharing one GPU (GPU test, RCCL refuses that).
"""
from __future__ import annotations

import numbers
import os
from typing import List, Optional, Protocol, Sequence, Union

import torch
import torch.distributed as dist


class GradWorker(Protocol):
    """What the driver needs from a per-rank worker (EdgeTrainer implements it on the GPU)."""

    def grad_step(self, view: int, wmap: torch.Tensor) -> torch.Tensor: ...
    def apply_adam(self, next_view: Optional[int] = None) -> None: ...


def init_from_env(backend: Optional[str] = None) -> tuple:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def librccl_path() -> str:
    """The RCCL PyTorch ships (torch/lib/librccl.so): dlopen'ed by the native data-parallel leg, so that the process
    holds one RCCL whether or not torch.distributed has loaded it already."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else "librccl.so"


def init_native_comm(group=None) -> int:
    """Creates the library's RCCL communicator for this process (eg_dp_init) from a ncclUniqueId that rank 0 draws and
    the ranks exchange over torch.distributed; a single process gets a one-rank communicator.  Returns its size.
    Idempotent.  (The communicator is separate from torch.distributed's own: the native run enqueues its collectives
    itself, on the launch stream.)"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > 1 and dist.get_backend(group) != "nccl":
        # (gloo = the several-ranks-per-GPU test mode: RCCL refuses two ranks on one device, the Python driver runs)
        raise RuntimeError(f"the native communicator needs the RCCL backend, the process group runs {dist.get_backend(group)}")
    have = int(lib.eg_dp_world())
    if have == world:
        return have
    if have > 0:  # a communicator of another size (an earlier group): never reported as ready for this one
        lib.eg_dp_shutdown()
    path = librccl_path().encode()
    buf = (C.c_ubyte * 128)()
    # rank 0's verdict travels with the id: a failure there must raise on EVERY rank, not leave the others in the broadcast
    err = ""
    if rank == 0 and lib.eg_dp_unique_id(path, buf) != 0:
        err = lib.eg_last_error_string().decode()
    if world > 1:
        t = torch.tensor([0 if not err else 1] + list(buf), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=0, group=group)
        vals = t.cpu().tolist()
        if vals[0] != 0:
            raise RuntimeError("eg_dp_unique_id failed on rank 0" + (f": {err}" if err else ""))
        buf = (C.c_ubyte * 128)(*vals[1:])
    elif err:
        raise RuntimeError(err)
    ok = lib.eg_dp_init(path, buf, rank, world) == 0
    msg = "" if ok else lib.eg_last_error_string().decode()
    if world > 1:  # ... and so must a failure of ncclCommInitRank on any rank
        t = torch.tensor([int(ok)], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        if int(t.item()) == 0:
            if ok:
                lib.eg_dp_shutdown()
            raise RuntimeError("eg_dp_init failed on " + ("this rank: " + msg if not ok else "another rank"))
    elif not ok:
        raise RuntimeError(msg)
    return world


def view_for(step: int, rank: int, world: int, n_views: int, views_per_rank: int = 1, slot: int = 0) -> int:
    """Round-robin view sharding: the step's batch is views {(step*world + r) * C + slot}, r = 0..world-1."""
    return ((step * world + rank) * views_per_rank + slot) % n_views


class DataParallelStep:
    def __init__(self, worker: GradWorker, group=None, time_comm: bool = False):
        self.worker = worker
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.time_comm = time_comm
        self.skip_collective = False  # measurement only: see no_collective()
        self._ev: List = []
        # EdgeTrainer keeps the journal: (grad_step, all-reduce, apply_adam) triples are journalled like its own steps,
        # its read-backs merge the sticky overflow / missed-stop words over the ranks (max) so that ALL ranks replay the
        # same steps, and the loss sums it reports are sums over the ranks
        self._journals = hasattr(worker, "attach_dp")
        if self._journals:
            worker.attach_dp(self)

    def no_collective(self):
        """Context manager, MEASUREMENT ONLY (bench.py's `ms_per_step_no_collective`): inside it the gradient all-reduce of
        `step` / `steps` is left out on every rank -- the Python driver's torch.distributed call and the native run's
        ncclAllReduce (eg_dp_force_all_reduce(-1)) alike -- so that the same launch sequence is timed without the wire.
        Every rank then steps on its own view's gradient: the replicas DIVERGE, whatever follows is not the reference's
        data-parallel training any more.  The small control collectives (read-back words, loss sums) stay."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            from . import _lib
            lib = _lib.load()
            native = lib.eg_dp_world() >= 1
            self.skip_collective = True
            if native:
                lib.eg_dp_force_all_reduce(-1)
            try:
                yield self
            finally:
                self.skip_collective = False
                if native:
                    lib.eg_dp_force_all_reduce(0)
        return cm()

    def reduce_words(self, ints, floats):
        """(element-wise max of `ints`, element-wise sum of `floats`) over the ranks, as Python lists."""
        if self.world <= 1 or not dist.is_initialized():
            return list(ints), list(floats)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        out_i, out_f = list(ints), list(floats)
        if ints:
            t = torch.tensor(ints, dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            out_i = [int(x) for x in t.tolist()]
        if floats:
            t = torch.tensor(floats, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            out_f = [float(x) for x in t.tolist()]
        return out_i, out_f

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        """In place from rank `src` (staged through the host in the several-ranks-on-one-GPU test mode)."""
        if self.world <= 1 or not dist.is_initialized():
            return t
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            host = t.cpu()
            dist.broadcast(host, src=src, group=self.group)
            if self.rank != src:
                t.copy_(host)
            return t
        dist.broadcast(t, src=src, group=self.group)
        return t

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks (what EdgeTrainer uses for the regulariser's running loss sum)."""
        self._all_reduce(t)
        return t

    # ------------------------------------------------------------------ the collective
    def _all_reduce(self, t: torch.Tensor, async_op: bool = False, gradient: bool = False):
        if self.world <= 1 or (gradient and self.skip_collective):
            return None
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            # test mode only (several ranks sharing one GPU, RCCL refuses that): stage through the host
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_reduce_scalar(self, x: float) -> float:
        """Sum of a host scalar over the ranks (the epoch's loss sum feeds the regulariser weights,
        train_gaussians.py:113,125: every replica must use the same value)."""
        if self.world <= 1 or not dist.is_initialized():
            return x
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def _mark(self):
        if self.time_comm and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return None

    # ------------------------------------------------------------------ one optimizer step
    def step(self, view: Union[int, Sequence[int]], wmap, next_view: Optional[int] = None) -> None:
        """`view`, `wmap`: this rank's view (+ weight map) of the step -- or lists of C views / maps, which run
        as batched launch sequences with the first half's all-reduce hidden behind the second half.
        next_view (single-view form): the view this rank takes in the NEXT step, when the caller knows it -- the
        post-reduce Adam then also projects + bins that view (EdgeTrainer.apply_adam(next_view): one launch instead
        of two, the tail fusion of the single-GPU step)."""
        if isinstance(view, numbers.Integral) or (isinstance(view, torch.Tensor) and view.dim() == 0):
            view = int(view)
        else:
            view, wmap = list(view), list(wmap)
        if self._journals:
            w = self.worker
            w._journal_push(("d", view, (wmap, next_view), w.epoch, w.loss_scale))
        self._step_raw(view, wmap, next_view)

    # ------------------------------------------------------------------ K optimizer steps by one native call
    def native_ready(self) -> bool:
        """The worker is an EdgeTrainer on the segmented layout and the library holds an RCCL communicator of this
        group's size (init_native_comm)."""
        from . import _lib
        w = self.worker
        return (hasattr(w, "_dp_steps_raw") and getattr(w, "seg_cap", 0) > 0
                and _lib.load().eg_dp_world() == (dist.get_world_size(self.group) if dist.is_initialized() else 1))

    def steps(self, views: Sequence[int], wmaps: Sequence[torch.Tensor], next_view: Optional[int] = None) -> None:
        """len(views) consecutive single-view steps of this rank (exactly `step(view, wmap, next_view=views[k + 1])` in
        a loop) enqueued by ONE native call: per step grad -> ncclAllReduce on the launch stream -> Adam + next
        projection (eg_train_steps_dp).  Journalled like the single steps (a replay runs them one by one through the
        same kernels).  Falls back to the loop when the native leg is not available (gloo, no communicator)."""
        views = [int(v) for v in views]
        nxt = views[1:] + [next_view]
        if not self.native_ready():
            for v, w_, n_ in zip(views, wmaps, nxt):
                self.step(v, w_, next_view=n_)
            return
        w = self.worker
        if self._journals:
            w._reserve_tags(len(views))
            for v, w_, n_ in zip(views, wmaps, nxt):
                w._journal_push(("d", v, (w_, n_), w.epoch, w.loss_scale), reserve=False)
        w._dp_steps_raw(views, list(wmaps), next_view, journalled=self._journals and w.replay_on_overflow)

    def _step_raw(self, view, wmap, next_view=None) -> None:
        """One (grad_step, all-reduce, apply_adam) triple -- also what a collective replay runs."""
        j = {"journalled": True} if self._journals and self.worker.replay_on_overflow else {}
        if isinstance(view, int):
            grads = self.worker.grad_step(view, wmap, **j)
            e0 = self._mark()
            self._all_reduce(grads, gradient=True)
            e1 = self._mark()
            if e0 is not None:
                self._ev.append((e0, e1))
            if next_view is None:
                self.worker.apply_adam()
            else:
                self.worker.apply_adam(next_view=int(next_view))
            return
        views, wmaps = list(view), list(wmap)
        if len(views) == 1 or self.world <= 1:
            grads = self.worker.grad_step_batched(views, wmaps, **j)
            e0 = self._mark()
            self._all_reduce(grads, gradient=True)
            e1 = self._mark()
            if e0 is not None:
                self._ev.append((e0, e1))
            self.worker.apply_adam()
            return
        h = (len(views) + 1) // 2  # odd C: the larger half first, its collective is the hidden one
        ga = self.worker.grad_step_batched(views[:h], wmaps[:h], slot=1, **j)   # first half -> second buffer
        work = self._all_reduce(ga, async_op=True, gradient=True)                              # ... reduced on RCCL's stream while
        gb = self.worker.grad_step_batched(views[h:], wmaps[h:], slot=0, **j)   # the second half is rasterised
        e0 = self._mark()
        self._all_reduce(gb, gradient=True)
        if work is not None:
            work.wait()  # the compute stream waits (no host block)
        e1 = self._mark()
        if e0 is not None:
            self._ev.append((e0, e1))
        gb.add_(ga)
        self.worker.apply_adam()

    def comm_us(self) -> Optional[float]:
        """Mean EXPOSED all-reduce time per step on this rank's compute stream, microseconds (time_comm=True)."""
        if not self._ev:
            return None
        torch.cuda.synchronize()
        us = [1e3 * a.elapsed_time(b) for a, b in self._ev]
        self._ev = []
        return sum(us) / len(us)
