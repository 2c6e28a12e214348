"""On-disk hand-off formats of the reference, so that the untouched downstream
stages (`fit_edges.py`, `eval.py`) consume this framework's output unchanged.

* PLY: exactly the vertex element the reference writes with `plyfile`
  (`edgegaussians/utils/io_utils.py:4-25`, called from `edge_gs.py:635-642`): little-endian binary,
  properties x y z scale1..3 quat1..4 opacity, all f4; scales are post-exp, quaternions wxyz as stored
  (un-normalised), opacity post-sigmoid.  `plyfile` is not needed to write or read it.
* state dict: the 4 tensors keyed like the reference's checkpoint (`edge_gs.py:625-633`,
  `train_utils.py:68-75`).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

_PLY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                       ("scale1", "<f4"), ("scale2", "<f4"), ("scale3", "<f4"),
                       ("quat1", "<f4"), ("quat2", "<f4"), ("quat3", "<f4"), ("quat4", "<f4"),
                       ("opacity", "<f4")])


def write_gaussian_params_as_ply(means, scales, quats, opacities, ply_path: str) -> None:
    """Same signature and file contents as io_utils.py:4-25 (arrays [N,3], [N,3], [N,4], [N,1])."""
    means, scales, quats, opacities = (np.asarray(a, dtype=np.float32) for a in (means, scales, quats, opacities))
    n = means.shape[0]
    v = np.zeros(n, dtype=_PLY_DTYPE)
    v["x"], v["y"], v["z"] = means[:, 0], means[:, 1], means[:, 2]
    v["scale1"], v["scale2"], v["scale3"] = scales[:, 0], scales[:, 1], scales[:, 2]
    v["quat1"], v["quat2"], v["quat3"], v["quat4"] = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
    v["opacity"] = opacities.reshape(n, -1)[:, 0]
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {n}\n" + \
        "".join(f"property float {name}\n" for name in _PLY_DTYPE.names) + "end_header\n"
    with open(ply_path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_gaussian_params_from_ply(ply_path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Same returns as io_utils.py:29-39: pos [N,3], scales [N,3], quats [N,4], opacities [N,1]."""
    with open(ply_path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, n, props = None, None, []
        while True:
            line = f.readline().decode("ascii").strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element" and tok[1] == "vertex":
                n = int(tok[2])
            elif tok[0] == "property":
                props.append((tok[2], {"float": "f4", "float32": "f4", "double": "f8"}[tok[1]]))
        if fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, max_rows=n).reshape(n, len(props))
            cols = {name: data[:, i].astype(np.float32) for i, (name, _) in enumerate(props)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            rec = np.frombuffer(f.read(), dtype=np.dtype([(nm, end + t) for nm, t in props]), count=n)
            cols = {nm: rec[nm].astype(np.float32) for nm, _ in props}
    pos = np.stack([cols["x"], cols["y"], cols["z"]], axis=1)
    scales = np.stack([cols["scale1"], cols["scale2"], cols["scale3"]], axis=1)
    quats = np.stack([cols["quat1"], cols["quat2"], cols["quat3"], cols["quat4"]], axis=1)
    return pos, scales, quats, cols["opacity"][:, None]


def export_as_ply(state: Dict[str, torch.Tensor], ply_path: str) -> None:
    """`EdgeGaussianSplatting.export_as_ply` (edge_gs.py:635-642) from a reference-keyed state dict."""
    write_gaussian_params_as_ply(
        state["gauss_params.means"].detach().cpu().numpy(),
        torch.exp(state["gauss_params.scales"]).detach().cpu().numpy(),
        state["gauss_params.quats"].detach().cpu().numpy(),
        torch.sigmoid(state["gauss_params.opacities"]).detach().cpu().numpy(), ply_path)
