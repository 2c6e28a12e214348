"""View-sharded data parallelism: one process per GPU, RCCL all-reduce of Gaussian gradients.

The reference is single-process / single-GPU (no torch.distributed call site) and
steps the optimizer after every single view (train_gaussians.py:104-106,311).  The path shards
naturally over VIEWS: every rank holds the full Gaussian state (11 floats/Gaussian +
Adam moments), takes a different view of the step's view batch, and the only exchange is ONE
all-reduce(sum) over a single fused [N,12] fp32 buffer per step
    [ dL/dmeans 3 | dL/dquats 4 | dL/dlog_scales 3 | dL/dlogit_opacity 1 | absgrad increment 1 ]
after which every rank applies the identical fused Adam -- replicas stay bit-identical because the
all-reduce result is identical on every rank.  This is a THROUGHPUT mode: a P-view batch per
optimizer step is a different trajectory from P sequential steps; what is guaranteed (and tested)
is  all-reduced gradient == sum of the single-GPU per-view gradients at the same parameters.

Backend "nccl" IS RCCL on ROCm; on the MI355X xGMI mesh a 1.4-24 MB buffer (N = 30k-500k) is
latency- to per-link-bound, so it is sent as one collective rather than per-parameter buckets.
The same code runs under "gloo" on CPU tensors (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Protocol

import torch
import torch.distributed as dist


class GradWorker(Protocol):
    """What the driver needs from a per-rank worker (EdgeTrainer implements it on the GPU)."""

    def grad_step(self, view: int, wmap: torch.Tensor) -> torch.Tensor: ...
    def apply_adam(self) -> None: ...


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


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """Round-robin view sharding: the step's batch is views {step*world + r}, r = 0..world-1."""
    return (step * world + rank) % n_views


class DataParallelStep:
    def __init__(self, worker: GradWorker, group=None):
        self.worker = worker
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self, view: int, wmap: torch.Tensor) -> None:
        grads = self.worker.grad_step(view, wmap)
        if self.world > 1:
            if grads.is_cuda and dist.get_backend(self.group) == "gloo":
                # test mode only (several ranks sharing one GPU, RCCL refuses that): stage through the host
                host = grads.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                grads.copy_(host)
            else:
                dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group)
        self.worker.apply_adam()
