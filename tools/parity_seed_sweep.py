"""Strict fused-step-vs-C-oracle parity (tests/util.py: check_fused_step_vs_c_oracle, 1e-4 on every element outside
the quantified borderline sets) on seeds / views / loss strategies / sizes OTHER than the ones the test-suite pins:
guards against a suite that passes by luck of its seed.  Prints one line per case; needs an MI355X."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegaussians_amd import _lib, synth
from tests.util import check_fused_step_vs_c_oracle
_lib.load()
cases = [("config1", 30000, 512, 512, s, v, st) for s, v, st in ((1, 0, "whole"), (2, 1, "weighted"), (3, 0, "bg_edge_ratio"), (4, 1, "whole"))]
cases += [("config3", 200000, 1600, 1200, 1, 0, "whole"), ("config4", 500000, 1200, 680, 2, 1, "weighted")]
# the headline scene (config 2 on the scan's real poses) on other seeds and views than the suite's
REAL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cameras_00004926.npz")
for seed, view, strat, spread, spec in ((1, 3, "weighted", True, False), (2, 44, "bg_edge_ratio", True, False),
                                        (3, 25, "whole", False, True), (4, 12, "whole", True, False)):
    try:
        sc = synth.make_scene(100000, 50, 512, 512, seed=seed, anisotropy=5.0, spread_opacity=spread, cameras_npz=REAL)
        check_fused_step_vs_c_oracle(sc, view, strat, f"sweep_config2_real_s{seed}_v{view}", speculate=spec)
        print("OK  ", "config2 real poses seed", seed, "view", view, strat, "spread", spread, "speculative", spec, flush=True)
    except AssertionError as e:
        print("FAIL", "config2 real poses seed", seed, "view", view, strat, "spread", spread, str(e)[:200], flush=True)
for name, n, W, H, seed, view, strat in cases:
    for spread in (True, False):
        try:
            sc = synth.make_scene(n, 2, W, H, seed=seed, anisotropy=5.0, spread_opacity=spread)
            check_fused_step_vs_c_oracle(sc, view, strat, f"sweep_{name}_s{seed}")
            print("OK  ", name, "seed", seed, "view", view, strat, "spread", spread, flush=True)
        except AssertionError as e:
            print("FAIL", name, "seed", seed, "view", view, strat, "spread", spread, str(e)[:200], flush=True)
