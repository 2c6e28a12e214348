"""Drop-in shim: ``from gsplat import rasterization`` (reference edgegaussians/models/edge_gs.py:8)
resolves to the MI355X-native implementation, so the reference's model class runs unchanged.

Only the one symbol the reference imports is provided."""
from edgegaussians_amd.rasterizer import rasterization  # noqa: F401

__version__ = "1.0.0+edgegaussians_amd"
__all__ = ["rasterization"]
