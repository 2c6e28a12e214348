#!/usr/bin/env python3
"""Copies the judged artefacts of an evidence run (tools/run_gpu_suite.sh -> gpurun_out/ev/) into profiles/ under
round-tagged names and derives the issue-side ("second") roofline of every kernel from the SQ counter pass:

    VALU issue rate = SQ_INSTS_VALU per launch / kernel duration, against the chip's wave64 VALU issue peak
                      (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction = 1229 G wave-instructions/s; round 2 priced 4 cycles)
    where the wave cycles go: SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY as shares of SQ_WAVE_CYCLES

usage: collect_profiles.py [r02]"""
import ast
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "ev")
OUT = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0  # (2 cycles per wave64 instruction: MI355X_MICROARCH.md, tools/microbench/issue_rates.hip)


def kernel_avgs(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        parts = line.split()
        if len(parts) < 7:
            continue
        name = " ".join(parts[:-6]).split("<")[0]
        calls, total = int(parts[-6]), float(parts[-5])
        a = out.setdefault(name, [0, 0.0])
        a[0] += calls
        a[1] += total
    return {k: v[1] / v[0] for k, v in out.items() if v[0]}


def main():
    copies = {"pytest_gpu.log": "pytest_gpu.log", "parity_report.jsonl": "parity_report.jsonl",
              "bench_default.json": "bench_default_config2.json", "bench_config1.json": "bench_config1.json",
              "bench_config3.json": "bench_config3.json", "bench_config4.json": "bench_config4.json",
              "bench_config2_init_opacity.json": "bench_config2_init_opacity.json",
              "bench_config3_init_opacity.json": "bench_config3_init_opacity.json",
              "bench_config4_init_opacity.json": "bench_config4_init_opacity.json",
              "issue_rates.txt": "microbench_issue_rates.txt", "operator_profile_config2.txt": "operator_profile_config2.txt", "fwd_wave_phases_config2.txt": "fwd_wave_phases_config2.txt",
              "bench_config2_force_dp.json": "bench_config2_force_dp.json", "regularizers_timing.json": "regularizers_timing.json",
              "train_abc_fixture.txt": "train_abc_fixture.txt", "late_epoch_bench.txt": "late_epoch_bench.txt",
              "bench_config2_operator_torch_adam.json": "bench_config2_operator_torch_adam.json",
              "bench_config2_operator_native_adam.json": "bench_config2_operator_native_adam.json"}
    copies.update({"fwd_wave_phases_config4.txt": "fwd_wave_phases_config4.txt", "roctx_ranges_config2.txt": "roctx_ranges_config2.txt",
                   "bench_config1_scenes1.json": "bench_config1_scenes1.json", "bench_config1_scenes2.json": "bench_config1_scenes2.json",
                   "bench_config1_scenes4.json": "bench_config1_scenes4.json", "bench_config1_scenes8.json": "bench_config1_scenes8.json",
                   "bench_config1_scenes8_threads.json": "bench_config1_scenes8_threads.json",
                   "bench_config2_scenes4.json": "bench_config2_scenes4.json"})
    copies.update({"bench_abc800.json": "bench_abc800_real_edges.json", "bench_driver_window.json": "bench_driver_window_20_steps.json"})
    for c in ("config1", "config2", "config2i", "config3", "config4", "abc800"):
        copies[f"kernel_stats_{c}.txt"] = f"kernel_stats_{c}.txt"
        copies[f"timeline_gaps_{c}.txt"] = f"timeline_gaps_{c}.txt"
        copies[f"sq_counters_{c}.txt"] = f"pmc_sq_counters_{c}.txt"
    for src, dst in copies.items():
        p = os.path.join(EV, src)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(OUT, f"{TAG}_{dst}"))
    for c in ("config1", "config2", "config2i"):
        ks, sq = os.path.join(EV, f"kernel_stats_{c}.txt"), os.path.join(EV, f"sq_counters_{c}.txt")
        if not (os.path.exists(ks) and os.path.exists(sq)):
            continue
        dur = kernel_avgs(ks)
        rows = {}
        for line in open(sq).read().splitlines():
            name, _, rest = line.partition(" {")
            if name.startswith("rocprim") or not rest:
                continue
            cnt = ast.literal_eval("{" + rest)
            us = dur.get(name)
            wc = max(cnt.get("SQ_WAVE_CYCLES", 0), 1)
            rows[name] = {
                "avg_launch_us": us,
                "valu_wave_instructions_per_launch": cnt.get("SQ_INSTS_VALU"),
                "valu_issue_G_per_s": (cnt.get("SQ_INSTS_VALU", 0) / (us * 1e-6) / 1e9) if us else None,
                "valu_issue_frac_of_peak": (cnt.get("SQ_INSTS_VALU", 0) / (us * 1e-6) / VALU_PEAK) if us else None,
                "share_of_wave_cycles": {"issuing": cnt.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                         "issuing_valu": cnt.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                         "parked_waitcnt_or_barrier": cnt.get("SQ_WAIT_ANY", 0) / wc,
                                         "issue_stalled": cnt.get("SQ_WAIT_INST_ANY", 0) / wc},
                "waves_per_launch": cnt.get("SQ_WAVES"),
            }
        json.dump({"valu_issue_peak_G_wave_instr_per_s": VALU_PEAK / 1e9,
                   "note": "SQ counters: one rocprofv3 --pmc pass (8 SQ slots), averages per launch; durations: the "
                           "--kernel-trace --stats pass of the same command; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count "
                           "quad-cycles summed over waves (MI355X_MICROARCH.md), so only their ratios are used",
                   "kernels": rows}, open(os.path.join(OUT, f"{TAG}_issue_roofline_{c}.json"), "w"), indent=1)
        for k, r in sorted(rows.items(), key=lambda kv: -(kv[1]["avg_launch_us"] or 0)):
            if r["avg_launch_us"]:
                s = r["share_of_wave_cycles"]
                print(f"{c} {k:34s} {r['avg_launch_us']:7.2f} us  VALU issue {100 * r['valu_issue_frac_of_peak']:5.1f} % of peak  "
                      f"issuing {100 * s['issuing']:4.1f} %  parked {100 * s['parked_waitcnt_or_barrier']:4.1f} %  stalled {100 * s['issue_stalled']:4.1f} %")


if __name__ == "__main__":
    main()
