"""Drop-in for the reference's optimizers: `edgegaussians_amd.optim.Adam` in place of `torch.optim.Adam`.

The reference keeps FOUR single-tensor `torch.optim.Adam` instances (train_utils.py:50-60: means, scales, quats,
opacities; betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad) and steps them one after the other after every
backward (train_gaussians.py:104-106, and 116-118 / 128-130 for the regulariser iterations).  On this host a
`torch.optim.Adam.step()` costs ~0.11 ms of Python and dispatch per call -- 0.44 ms per training step, half of what the
whole drop-in protocol costs here -- for an update that takes the GPU 2 us.  This class keeps torch's interface and
STATE LAYOUT (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`, `param_groups[i]["lr"]`), so the schedulers of
train_utils.py and the model's densification code, which edits `optimizer.state` and `param_groups[0]["params"]` in
place (edge_gs.py:384-402, 431-451), work unchanged; `step()` is one native launch per tensor (`eg_adam_tensor`, the
arithmetic of `eg_adam_multi`).

Not supported (the reference uses none of them): weight decay, amsgrad, maximize, closures that re-evaluate the loss,
sparse gradients, non-fp32 or non-contiguous parameters, step pre / post hooks -- each raises instead of falling back.

Rounding: the native kernel forms 1 / (sqrt(v_hat) + eps) with the hardware square root and reciprocal (1 ulp each),
torch's with IEEE sqrt and a division: the two agree to ~2e-7 relative per step (tests: 2e-6 over 31 steps incl. the
reference's in-place state surgery), NOT bit for bit.
"""
from __future__ import annotations

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("edgegaussians_amd.optim.Adam: weight_decay / amsgrad are not implemented "
                                      "(the reference uses neither: train_utils.py:50-60)")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    # (no torch.no_grad(): nothing below is recorded by autograd -- the update runs on raw device pointers)
    def step(self, closure=None, zero_grad: bool = False):
        """One Adam step on every parameter that has a gradient.  `zero_grad=True` clears the gradients in the same
        launch (what `opt.step(); opt.zero_grad(set_to_none=False)` does in two)."""
        if closure is not None:
            raise NotImplementedError("edgegaussians_amd.optim.Adam.step: closures are not supported")
        _lib.load()  # fails loudly without the HIP library: there is no torch fallback
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_cuda:
                    raise NotImplementedError("edgegaussians_amd.optim.Adam: dense fp32 device tensors only")
                if not (p.is_contiguous() and g.is_contiguous()):
                    raise NotImplementedError("edgegaussians_amd.optim.Adam: contiguous parameters and gradients only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)  # torch's layout: a CPU scalar tensor
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.shape != p.shape or v.shape != p.shape or not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("edgegaussians_amd.optim.Adam: optimizer state does not match its parameter")
                _lib.call("eg_adam_tensor", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                          float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]),
                          1 if zero_grad else 0, _lib.stream())
        return None

    # torch.optim.Optimizer wraps `step` of every subclass in a profiler range + pre/post-hook dispatch unless it is
    # marked as hooked already: ~25 us of host time per call, four calls per training step.  This class has no step hooks.
    step.hooked = True

    def register_step_pre_hook(self, hook):
        raise NotImplementedError("edgegaussians_amd.optim.Adam: step hooks are not dispatched (step() skips torch's "
                                  "hook wrapper); use torch.optim.Adam if you need them")

    def register_step_post_hook(self, hook):
        raise NotImplementedError("edgegaussians_amd.optim.Adam: step hooks are not dispatched (step() skips torch's "
                                  "hook wrapper); use torch.optim.Adam if you need them")

    def zero_grad(self, set_to_none: bool = True):
        """torch's semantics (gradients dropped, or zeroed in place with set_to_none=False) without its profiler range
        and foreach grouping: ~3 us instead of ~20 per call."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.requires_grad_(False)
                    p.grad.zero_()
