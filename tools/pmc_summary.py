#!/usr/bin/env python3
"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as
MI355X_MICROARCH.md 'HBM' prescribes: the counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream at 64 B, so the read side is reported both raw and doubled
(the doubled figure is the calibrated one for 16 B/lane streaming reads; gather-heavy kernels lie in
between); WRITE_SIZE is uncalibrated.  Working sets below ~100 MB sit in the 256 MB Infinity Cache,
whose hits are counted too, so these are L2-miss ("fabric") bytes rather than DRAM bytes.

usage: pmc_summary.py <fetch.db> <write.db> <out.json>
"""
import json
import sqlite3
import sys

STAGE_OF = {
    "project_fwd_kernel": "project_bin", "project_emit_kernel": "project_bin", "tile_offsets_kernel": "tile_offsets", "tile_emit_kernel": "tile_emit",
    "tile_sort_kernel": "tile_sort", "composite_slice_fwd_kernel": "composite_slice_fwd",
    "composite_combine_fwd_kernel": "composite_combine_fwd", "composite_rewalk_fwd_kernel": "composite_rewalk_fwd",
    "footprint_bwd_kernel": "footprint_bwd",
    "project_bwd_kernel": "project_bwd_adam",
}


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for name, v in rows:
        short = name.split("(")[0].replace("void ", "").replace("eg::", "").split("<")[0]
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels, stages = {}, {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fetch = 1024.0 * sf / max(nf, 1)
        write = 1024.0 * sw / max(nw, 1)
        kernels[k] = {"launches": nf, "fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write}
    # per stage: per-step totals (a stage may be two kernels / two variants launched once each)
    for k, d in kernels.items():
        st = STAGE_OF.get(k)
        if not st:
            continue
        s = stages.setdefault(st, {"fetch_bytes_raw": 0.0, "fetch_bytes_x2": 0.0, "write_bytes": 0.0})
        per_step = 1.0
        if k == "tile_sort_kernel":
            per_step = 2.0  # two variants per step, averaged above over both
        for key in s:
            s[key] += per_step * d[key]
    for st, s in stages.items():
        s["hbm_bytes_per_launch"] = s["fetch_bytes_x2"] + s["write_bytes"]
        s["hbm_bytes_per_launch_raw"] = s["fetch_bytes_raw"] + s["write_bytes"]
    out = dict(stages)
    out["_kernels"] = kernels
    out["_note"] = ("FETCH_SIZE/WRITE_SIZE from separate rocprofv3 --pmc passes, KiB -> bytes; hbm_bytes_per_launch "
                    "uses the gfx950 x2 read correction of MI355X_MICROARCH.md; Infinity-Cache hits are included "
                    "(working set < 256 MB)")
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for st, s in sorted(stages.items()):
        print(f"{st:26s} fetch_raw {s['fetch_bytes_raw']/1e6:8.2f} MB  x2 {s['fetch_bytes_x2']/1e6:8.2f} MB  write {s['write_bytes']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
