"""ctypes front end of the plain-C oracle (oracle/eg_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`build()` compiles oracle/eg_oracle.c with gcc through oracle/Makefile; the product never links it.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libeg_oracle.so")
_lib: Optional[C.CDLL] = None

_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_d = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "eg_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.ego_isect_count.restype = C.c_int64
        lib.ego_train_step.restype = C.c_double
        lib.ego_num_threads.restype = C.c_int
        _lib = lib
    return _lib


def _opt(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def rasterize(means, quats, scales, opacities, colors, viewmat, K, width, height, near_plane=0.01,
              far_plane=1e10, eps2d=0.3, radius_clip=0.0, antialiased=True) -> Dict[str, np.ndarray]:
    """Forward of the reference's call for one camera; numpy fp32 in, dict of numpy out."""
    lib = load()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    means, quats, scales, opacities, colors = map(f32, (means, quats, scales, opacities, colors))
    vm, Kc = f32(viewmat).reshape(16), f32(K).reshape(9)
    N, CH = means.shape[0], colors.shape[1]
    tw, th = math.ceil(width / 16), math.ceil(height / 16)
    radii = np.zeros(N, np.int32)
    means2d, depths = np.zeros((N, 2), np.float32), np.zeros(N, np.float32)
    conics, comps = np.zeros((N, 3), np.float32), np.zeros(N, np.float32)
    lib.ego_project_fwd(_opt(means), _opt(quats), _opt(scales), _opt(vm), _opt(Kc), N, width, height,
                        C.c_float(near_plane), C.c_float(far_plane), C.c_float(eps2d), C.c_float(radius_clip),
                        _opt(radii), _opt(means2d), _opt(depths), _opt(conics), _opt(comps))
    opac = f32(opacities * comps) if antialiased else opacities
    tpg = np.zeros(N, np.int32)
    M = int(lib.ego_isect_count(_opt(means2d), _opt(radii), N, width, height, _opt(tpg)))
    ids, flat = np.zeros(max(M, 1), np.int64), np.zeros(max(M, 1), np.int32)
    offsets = np.zeros(tw * th, np.int32)
    lib.ego_isect_emit_sort(_opt(means2d), _opt(radii), _opt(depths), N, width, height, C.c_int64(M), _opt(ids),
                            _opt(flat), _opt(offsets))
    render = np.zeros((height, width, CH), np.float32)
    alphas = np.zeros((height, width), np.float32)
    last = np.zeros((height, width), np.int32)
    lib.ego_composite_fwd(_opt(means2d), _opt(conics), _opt(colors), _opt(opac), CH, width, height, _opt(offsets),
                          _opt(flat), C.c_int64(M), _opt(render), _opt(alphas), _opt(last))
    return dict(radii=radii, means2d=means2d, depths=depths, conics=conics, comps=comps, opacities=opac,
                tiles_per_gauss=tpg, isect_ids=ids[:M], flatten_ids=flat[:M], isect_offsets=offsets.reshape(th, tw),
                render=render, alphas=alphas, last_ids=last, M=M, _means=means, _quats=quats, _scales=scales,
                _op_in=opacities, _colors=colors, _vm=vm, _K=Kc, _size=(width, height), _eps2d=eps2d, _aa=antialiased)


def borderline_pixels(fw: Dict[str, np.ndarray], rel_alpha: float = 1e-4, rel_T: float = 2e-4) -> np.ndarray:
    """uint8 [H,W]: pixels whose walk passes within the margins of a float threshold (see eg_oracle.c)."""
    width, height = fw["_size"]
    flat = np.ascontiguousarray(np.concatenate([fw["flatten_ids"], np.zeros(1, np.int32)]))
    mask = np.zeros((height, width), np.uint8)
    load().ego_borderline_pixels(_opt(fw["means2d"]), _opt(fw["conics"]), _opt(fw["opacities"]), width, height,
                                 _opt(np.ascontiguousarray(fw["isect_offsets"].reshape(-1))), _opt(flat),
                                 C.c_int64(fw["M"]), C.c_double(rel_alpha), C.c_double(rel_T), _opt(mask))
    return mask


def stopped_pixels(fw: Dict[str, np.ndarray]) -> np.ndarray:
    """bool [H,W]: pixels whose front-to-back walk ended on the transmittance rule (next T <= 1e-4)."""
    width, height = fw["_size"]
    flat = np.ascontiguousarray(np.concatenate([fw["flatten_ids"], np.zeros(1, np.int32)]))
    out = np.zeros((height, width), np.uint8)
    load().ego_composite_stopped(_opt(fw["means2d"]), _opt(fw["conics"]), _opt(fw["opacities"]), width, height,
                                 _opt(np.ascontiguousarray(fw["isect_offsets"].reshape(-1))), _opt(flat),
                                 C.c_int64(fw["M"]), _opt(out))
    return out > 0


def project_borderline(means, quats, scales, viewmat, K, width, height, near_plane=0.01, eps2d=0.3,
                       rel: float = 2e-5) -> np.ndarray:
    """uint8 [N]: Gaussians whose integer decisions (radius ceil, culls, tile box) hinge on float rounding."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    means, quats, scales = map(f32, (means, quats, scales))
    mask = np.zeros(means.shape[0], np.uint8)
    load().ego_project_borderline(_opt(means), _opt(quats), _opt(scales), _opt(f32(viewmat).reshape(16)),
                                  _opt(f32(K).reshape(9)), means.shape[0], width, height, C.c_double(near_plane),
                                  C.c_double(eps2d), C.c_double(rel), _opt(mask))
    return mask


def backward(fw: Dict[str, np.ndarray], v_render: np.ndarray, v_alphas: Optional[np.ndarray] = None):
    """Gradients of the call w.r.t. means, quats, scales, opacities, colors (+ means2d grad / absgrad)."""
    lib = load()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    width, height = fw["_size"]
    N, CH = fw["_means"].shape[0], fw["_colors"].shape[1]
    M = fw["M"]
    flat = np.ascontiguousarray(np.concatenate([fw["flatten_ids"], np.zeros(1, np.int32)]))
    offsets = np.ascontiguousarray(fw["isect_offsets"].reshape(-1))
    v_render = f32(v_render)
    v_alphas = f32(v_alphas) if v_alphas is not None else None
    v_m2d, v_abs = np.zeros((N, 2), np.float32), np.zeros((N, 2), np.float32)
    v_con, v_col, v_op = np.zeros((N, 3), np.float32), np.zeros((N, CH), np.float32), np.zeros(N, np.float32)
    lib.ego_composite_bwd(_opt(fw["means2d"]), _opt(fw["conics"]), _opt(fw["_colors"]), _opt(fw["opacities"]), CH,
                          width, height, _opt(offsets), _opt(flat), C.c_int64(M), _opt(fw["alphas"]),
                          _opt(fw["last_ids"]), _opt(v_render), _opt(v_alphas), _opt(v_m2d), _opt(v_abs), _opt(v_con),
                          _opt(v_col), _opt(v_op))
    if fw["_aa"]:
        v_comp, v_opac = f32(v_op * fw["_op_in"]), v_op * fw["comps"]
    else:
        v_comp, v_opac = None, v_op
    g_means, g_quats, g_scales = np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32), np.zeros((N, 3), np.float32)
    lib.ego_project_bwd(_opt(fw["_means"]), _opt(fw["_quats"]), _opt(fw["_scales"]), _opt(fw["_vm"]), _opt(fw["_K"]), N,
                        width, height, C.c_float(fw["_eps2d"]), _opt(fw["radii"]), _opt(v_m2d), None, _opt(v_con),
                        _opt(v_comp), _opt(g_means), _opt(g_quats), _opt(g_scales))
    return dict(means=g_means, quats=g_quats, scales=g_scales, opacities=v_opac, colors=v_col, means2d=v_m2d,
                absgrad=v_abs, conics=v_con, opacities_eff=v_op)


class CpuTrainer:
    """The whole per-view training step in C (cpu_baseline of bench.py)."""

    def __init__(self, means, log_scales, quats, logit_opacities, lrs):
        f32 = lambda a: np.array(np.asarray(a), dtype=np.float32, order="C", copy=True)  # noqa: E731  (own the state)
        self.means, self.log_scales, self.quats = f32(means), f32(log_scales), f32(quats)
        self.logit = f32(logit_opacities).reshape(-1)
        self.N = self.means.shape[0]
        self.m, self.v = np.zeros(11 * self.N, np.float32), np.zeros(11 * self.N, np.float32)
        self.absgrads = np.zeros(self.N, np.float32)
        self.lr = np.ascontiguousarray([lrs["means"], lrs["scales"], lrs["quats"], lrs["opacities"]], dtype=np.float64)
        self.step = 0

    def train_step(self, viewmat, K, width, height, gt, wmap):
        f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)  # noqa: E731
        self.step += 1
        M = C.c_int64(0)
        loss = load().ego_train_step(_opt(self.means), _opt(self.log_scales), _opt(self.quats), _opt(self.logit),
                                     _opt(self.m), _opt(self.v), _opt(self.absgrads), self.N, _opt(f32(viewmat).reshape(16)),
                                     _opt(f32(K).reshape(9)), width, height, _opt(f32(gt)), _opt(f32(wmap)), _opt(self.lr),
                                     self.step, C.byref(M))
        return float(loss), int(M.value)


def num_threads() -> int:
    return int(load().ego_num_threads())
