"""edgegaussians_amd -- MI355X-native edge-Gaussian rasterizer (the hot path of
kunalchelani/EdgeGaussians) behind the reference's own operator surface.

    from edgegaussians_amd import rasterization      # == the reference's gsplat.rasterization call
    from edgegaussians_amd import EdgeTrainer        # fused per-view training step (train_gaussians.py:71-106)
"""
from .rasterizer import rasterization  # noqa: F401
from .trainer import EdgeTrainer, LRSchedule, train_steps_multi  # noqa: F401
from .train_loop import train, train_epoch  # noqa: F401

__all__ = ["rasterization", "EdgeTrainer", "LRSchedule", "train", "train_epoch", "train_steps_multi"]
