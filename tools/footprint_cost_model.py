"""Issue-slot cost model of the footprint backward's lane dealing, on the bench scenes (CPU only, no GPU).
Compares the shipped walk (stride dealing over the sheared box: ~57 slots per visited cell, ~400 per wave) with
line-run variants.  python tools/footprint_cost_model.py config2 [view]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegaussians_amd import synth  # noqa: E402
from oracle import ref_torch as O  # noqa: E402
import bench  # noqa: E402


def walks(name, view=0, spread=True):
    n, v, w, h = bench.CONFIGS[name]
    real = name in ("config1", "config2") and os.path.exists(bench.REAL_POSES)
    sc = synth.make_scene(n, min(v, 4), w, h, seed=0, anisotropy=5.0, spread_opacity=spread,
                          cameras_npz=bench.REAL_POSES if real else None)
    radii, m2d, depth, conic, comp = O.project(sc.means, sc.quats, torch.exp(sc.log_scales), sc.viewmats[view], sc.Ks[view], w, h)
    op = torch.sigmoid(sc.logit_opacities[:, 0]) * comp
    ok = (radii > 0) & (op * 255 > 1)
    a, b, c = (conic[ok, i].double().numpy() for i in range(3))
    x, y = m2d[ok, 0].double().numpy(), m2d[ok, 1].double().numpy()
    thr = np.log(255 * op[ok].double().numpy())
    det = a * c - b * b
    k = 2 * thr / det
    ex, ey = np.sqrt(k * c) * 1.001 + 0.01, np.sqrt(k * a) * 1.001 + 0.01
    j0 = np.maximum(0, np.ceil(x - ex - 0.5)); j1 = np.minimum(w - 1, np.floor(x + ex - 0.5))
    i0 = np.maximum(0, np.ceil(y - ey - 0.5)); i1 = np.minimum(h - 1, np.floor(y + ey - 0.5))
    fw, fh = j1 - j0 + 1, i1 - i0 + 1
    vis = (fw > 0) & (fh > 0)
    hw = np.sqrt(2 * thr / a) * 1.001 + 0.01
    hh = np.sqrt(2 * thr / c) * 1.001 + 0.01
    pw = np.minimum(np.floor(2 * hw) + 1, fw)      # row-sheared box: fh rows of pw cells
    ph = np.minimum(np.floor(2 * hh) + 1, fh)      # column-sheared box: fw columns of ph cells
    area = np.pi * 2 * thr / np.sqrt(det)           # pixels inside the ellipse (unclipped)
    return dict(pw=pw[vis], fh=fh[vis], ph=ph[vis], fw=fw[vis], area=area[vis], n_all=n, n_vis=int(vis.sum()))


def model_current(pw, fh, G=8, per_visit=57.0, per_wave=400.0):
    cells = (pw * fh).astype(np.int64)
    n = len(cells)
    pad = (-n) % G
    cells = np.concatenate([cells, np.zeros(pad, np.int64)]).reshape(-1, G)
    total = cells.sum(1)
    live = (cells > 0).sum(1)
    share = (64 - live)[:, None] / np.maximum(total, 1)[:, None]
    lanes = np.where(cells > 0, 1 + np.floor(cells * share), 0)
    slack = 64 - lanes.sum(1)
    rank = np.cumsum(cells > 0, 1) - (cells > 0)
    lanes = lanes + ((cells > 0) & (rank < slack[:, None]))
    per_lane = np.where(cells > 0, np.ceil(cells / np.maximum(lanes, 1)), 0)
    per_lane = 2 * np.ceil(per_lane / 2)  # two streams
    steps = per_lane.max(1)
    return float((steps * per_visit + per_wave).sum()), float(cells.sum())


def model_runs(ll, nl, G=8, L=8, per_cell=37.0, per_run=35.0, per_wave=450.0, sort_group=1):
    """Lines of length ll (nl of them per Gaussian) cut into runs of <= L cells; lanes of a wave dealt over its G
    Gaussians in proportion to their run counts; a wave step = run set-up + L visits."""
    runs = (np.ceil(ll / L) * nl).astype(np.int64)
    n = len(runs)
    order = np.arange(n)
    if sort_group > 1:  # Gaussians of sort_group consecutive waves ranked by line length
        blk = G * sort_group
        padn = (-n) % blk
        key = np.concatenate([ll, np.full(padn, 1e9)])
        order = (np.argsort(key.reshape(-1, blk), axis=1, kind="stable") + (np.arange(len(key) // blk) * blk)[:, None]).reshape(-1)
        order = order[order < n]
    runs = runs[order]
    pad = (-n) % G
    runs = np.concatenate([runs, np.zeros(pad, np.int64)]).reshape(-1, G)
    total = runs.sum(1)
    live = (runs > 0).sum(1)
    share = (64 - live)[:, None] / np.maximum(total, 1)[:, None]
    lanes = np.where(runs > 0, 1 + np.floor(runs * share), 0)
    slack = 64 - lanes.sum(1)
    rank = np.cumsum(runs > 0, 1) - (runs > 0)
    lanes = lanes + ((runs > 0) & (rank < slack[:, None]))
    steps = np.where(runs > 0, np.ceil(runs / np.maximum(lanes, 1)), 0).max(1)
    return float((steps * (per_run + L * per_cell) + per_wave).sum())


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "config2"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    d = walks(name, view)
    pw, fh, ph, fw = d["pw"], d["fh"], d["ph"], d["fw"]
    print(f"{name} view {view}: {d['n_vis']} of {d['n_all']} Gaussians with a footprint")
    for nm, v in (("pw", pw), ("fh", fh), ("ph", ph), ("fw", fw), ("cells row-sheared", pw * fh), ("ellipse area", d["area"])):
        q = np.percentile(v, [5, 25, 50, 75, 95])
        print(f"  {nm:18s} mean {v.mean():8.1f}  p5/25/50/75/95 {q}")
    cur, cells = model_current(pw, fh)
    print(f"  current: {cur / 1e6:.1f} M slots ({cur / cells:.1f} per cell of the row-sheared box; inside-ellipse fraction {d['area'].sum() / cells:.2f})")
    # long direction per Gaussian: rows of pw (fh lines) or columns of ph (fw lines), whichever line is longer
    rowwise = pw >= ph
    ll = np.where(rowwise, pw, ph)
    nl = np.where(rowwise, fh, fw)
    print(f"  lines along the long direction: length mean {ll.mean():.1f} p5/50/95 {np.percentile(ll, [5, 50, 95])}; lines per Gaussian {nl.mean():.1f}")
    for L in (4, 6, 8, 12, 16):
        for sg in (1, 4, 16):
            r = model_runs(ll, nl, L=L, sort_group=sg)
            r2 = model_runs(pw, fh, L=L, sort_group=sg)
            print(f"  runs L={L:2d} sort over {sg:2d} waves: long-direction {r / 1e6:7.1f} M slots ({r / cur:.2f}x)   rows only {r2 / 1e6:7.1f} ({r2 / cur:.2f}x)")
