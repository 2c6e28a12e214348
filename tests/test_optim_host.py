"""Host-side contract of the drop-in optimizer class (edgegaussians_amd/optim.py) -- what can be checked without a GPU:
it is a torch.optim.Optimizer with torch.optim.Adam's constructor surface (train_utils.py:50-60 builds the reference's
four optimizers), schedulers attach to it, unsupported options raise, and a step without the HIP library fails loudly
instead of falling back to torch arithmetic."""
import pytest
import torch

from edgegaussians_amd.optim import Adam


def test_constructor_surface_and_schedulers():
    p = torch.nn.Parameter(torch.zeros(8, 3))
    o = Adam([p], lr=2e-3)
    assert isinstance(o, torch.optim.Optimizer)
    g = o.param_groups[0]
    assert g["lr"] == 2e-3 and tuple(g["betas"]) == (0.9, 0.999) and g["eps"] == 1e-8  # torch.optim.Adam's defaults
    assert g["params"][0] is p and len(o.state) == 0
    sch = torch.optim.lr_scheduler.MultiStepLR(o, milestones=[1], gamma=0.1)  # train_utils.py:51
    with pytest.warns(UserWarning):  # (torch's own warning: scheduler stepped before the optimizer)
        sch.step()
    assert abs(o.param_groups[0]["lr"] - 2e-4) < 1e-12
    for bad in (dict(weight_decay=0.1), dict(amsgrad=True)):
        with pytest.raises(NotImplementedError):
            Adam([p], lr=1e-3, **bad)
    with pytest.raises(ValueError):
        Adam([p], lr=-1.0)


def test_zero_grad_semantics():
    p = torch.nn.Parameter(torch.zeros(5))
    o = Adam([p], lr=1e-3)
    p.grad = torch.ones(5)
    o.zero_grad(set_to_none=False)
    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0
    p.grad = torch.ones(5)
    o.zero_grad()
    assert p.grad is None


@pytest.mark.skipif(torch.cuda.is_available(), reason="the loud failure is the no-GPU behaviour")
def test_step_without_the_hip_library_raises():
    p = torch.nn.Parameter(torch.zeros(5))
    o = Adam([p], lr=1e-3)
    p.grad = torch.ones(5)
    with pytest.raises(RuntimeError):
        o.step()
    assert float(p.detach().abs().sum()) == 0.0  # nothing was updated by some fallback


def test_step_hooks_are_refused_not_silently_dropped():
    """step() skips torch's hook wrapper, so a registered hook would never fire -- registering one raises."""
    import pytest
    import torch
    from edgegaussians_amd.optim import Adam
    opt = Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    with pytest.raises(NotImplementedError):
        opt.register_step_pre_hook(lambda *a: None)
    with pytest.raises(NotImplementedError):
        opt.register_step_post_hook(lambda *a: None)
