"""Fused step against the plain-C oracle at the sizes BASELINE.json names (configs 1, 3, 4; config 2 and
config 5's per-GPU workload are the same shape and are covered by test_gpu_parity.py's config-2 tests).

One view per size: gradients of `eg_train_step` (Adam off) against the C oracle's forward + backward on the
same parameters, then one whole fused step (Adam on) against `ego_train_step`.  Floats 1e-4 on EVERY element:
the Gaussians whose integer decisions are float-borderline are taken out of the scene and the borderline
pixels get zero loss weight on both sides (tests/util.py: check_fused_step_vs_c_oracle); both sets are
counted and reported (gpurun_out/parity_report.jsonl -> profiles/).
"""
import pytest

from tests.util import check_fused_step_vs_c_oracle

pytestmark = pytest.mark.gpu

SIZES = {
    # name: (Gaussians, width, height, view, strategy)
    "config1": (30_000, 512, 512, 0, "whole"),
    "config3": (200_000, 1600, 1200, 1, "weighted"),
    "config4": (500_000, 1200, 680, 0, "bg_edge_ratio"),
}


@pytest.mark.parametrize("name", sorted(SIZES))
def test_fused_step_vs_c_oracle_full_size(name):
    from edgegaussians_amd import _lib, synth
    _lib.load()
    n, W, H, view, strategy = SIZES[name]
    sc = synth.make_scene(n, 2, W, H, seed=0, anisotropy=5.0, spread_opacity=True)
    check_fused_step_vs_c_oracle(sc, view, strategy, name)
