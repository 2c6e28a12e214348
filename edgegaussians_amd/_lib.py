"""ctypes binding of libedgegs.so (the C ABI declared in include/edgegs.h).

There is NO fallback: if the shared library is missing or no gfx950 device is visible, the
product raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never imported
from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libedgegs.so")

_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_u32 = C.c_uint32
_f = C.c_float

FLAG_LOG_SCALES = 1
FLAG_LOGIT_OPACITIES = 2
FLAG_ANTIALIASED = 4
FLAG_TIGHT_TILES = 8
FLAG_ABSGRAD_WRITE = 16
FLAG_FRONT_PREFIX = 32   # EG_FLAG_FRONT_PREFIX (tile grids above PREFIX_HERE_MAX_TILES: `ticket` is [T + 2])
PREFIX_HERE_MAX_TILES = 2560  # kPrefixHereMaxTiles (csrc/common.h)
REWALK_SPECULATE = -2  # EG_REWALK_SPECULATE
MAX_WS_TAG = 0xfffe      # EG_MAX_WS_TAG


class AdamHyper(C.Structure):
    _fields_ = [("lr_means", C.c_double), ("lr_scales", C.c_double), ("lr_quats", C.c_double),
                ("lr_opacities", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("step", _i32), ("group_steps", _i32 * 4)]


class StepArgs(C.Structure):
    _fields_ = [
        ("means", _vp), ("quats", _vp), ("log_scales", _vp), ("logit_opacities", _vp),
        ("adam_m", _vp), ("adam_v", _vp), ("absgrads", _vp), ("N", _i32),
        ("viewmat", _vp), ("K", _vp), ("gt", _vp), ("wmap", _vp),
        ("width", _i32), ("height", _i32), ("loss_scale", _f),
        ("splat", _vp), ("g2d", _vp),
        ("tile_counts", _vp), ("offsets", _vp), ("item_offsets", _vp), ("total", _vp),
        ("tile_mask", _vp), ("ticket", _vp),
        ("workspace", _vp), ("max_items", _i64),
        ("keys", _vp), ("flatten_ids", _vp), ("capacity", _i64), ("max_tile_hint", _i32), ("rewalk_hint", _i32),
        ("seg_cap", _i32), ("tile_end", _vp), ("item_end", _vp), ("item_tile", _vp),
        ("render", _vp), ("alphas", _vp), ("vpix", _vp), ("loss", _vp), ("gtstop", _vp),
        ("last_ids", _vp),
        ("v_means", _vp), ("v_quats", _vp), ("v_scales", _vp), ("v_opacities", _vp),
        ("adam_host", C.POINTER(AdamHyper)),
        ("next_viewmat", _vp), ("next_K", _vp), ("have_projection", _i32), ("ws_tag", _i32),
        ("item_rec", _vp), ("two_kernel_backward", _i32),
    ]


class OperatorArgs(C.Structure):
    """eg_operator_args (include/edgegs.h): the drop-in operator's fast path."""
    _fields_ = [
        ("means", _vp), ("quats", _vp), ("scales", _vp), ("opacities", _vp), ("colors", _vp), ("color_channels", _i32),
        ("viewmat", _vp), ("K", _vp), ("N", _i32), ("width", _i32), ("height", _i32), ("flags", _u32),
        ("splat", _vp), ("alphas", _vp), ("means2d", _vp), ("gtstop", _vp),
        ("tile_counts", _vp), ("tile_start", _vp), ("tile_end", _vp), ("item_first", _vp), ("item_end", _vp),
        ("item_tile", _vp), ("item_rec", _vp), ("total", _vp), ("ticket", _vp), ("keys", _vp), ("flatten_ids", _vp),
        ("seg_cap", _i32), ("max_tile_hint", _i32), ("max_items", _i64), ("workspace", _vp), ("ws_tag", _i32),
    ]


# name -> argtypes; every entry returns int and ends with the stream
_SIGS = {
    "eg_project_fwd": [_vp] * 6 + [_i32, _i32, _i32, _f, _f, _f, _f, _u32] + [_vp] * 9 + [_vp],
    "eg_tile_count": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "eg_tile_offsets": [_vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "eg_tile_emit": [_vp, _vp, _vp, _vp, _u32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp],
    "eg_project_bin": [_vp] * 6 + [_i32, _i32, _i32, _u32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "eg_sort_pairs": [_vp, _vp, _i32, _i64, _vp, _vp, _i32, _vp],
    "eg_project_emit": [_vp] * 6 + [_i32, _i32, _i32, _u32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp],
    "eg_sort_segments": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "eg_composite_fwd_segments": [_vp] * 7 + [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp],
    "eg_composite_fwd": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                         _vp, _vp, _i64, _vp, _vp, _i32, _vp],
    "eg_composite_bwd": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "eg_composite_bwd_colors": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eg_project_bwd": [_vp] * 6 + [_i32, _i32, _i32, _f, _u32] + [_vp] * 9 + [_vp],
    "eg_absgrad_accum": [_vp, _i32, _vp, _vp],
    "eg_composite_bwd_footprint": [_vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "eg_backward_fused": [_vp] * 6 + [_i32, _i32, _i32, _f, _u32] + [_vp] * 10 + [C.POINTER(AdamHyper), _vp],
    "eg_adam_multi": [_vp] * 10 + [_i32, AdamHyper, _vp, _vp, _vp],
    "eg_adam_tensor": [_vp] * 4 + [_i64, C.c_double, C.c_double, C.c_double, C.c_double, _i32, _i32, _vp],
    "eg_adam_emit": [_vp] * 10 + [_i32, AdamHyper, _vp, _vp, _vp, _vp, _i32, _i32, _u32, _vp, _vp, _i32, _vp, _vp, _i32,
                     _vp, _vp, _vp],
    "eg_project_bwd_adam": [_vp] * 6 + [_i32, _i32, _i32, _f, _u32] + [_vp] * 5 + [AdamHyper, _vp],
    "eg_mask_scan": [_vp, _i32, _vp, _vp, _vp],
    "eg_compact_rows": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "eg_append_rows": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _f, _vp, _vp],
    "eg_ratio_wmap": [_vp, _f, _i32, _vp, _i32, _i32, _vp, _vp],
    "eg_ratio_wmap_seeded": [_vp, _f, _i32, _i32, _i32, C.c_uint64, _i32, _vp, _vp],
    "eg_ratio_wmaps_seeded": [_i32, C.POINTER(_vp), _f, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_uint64),
                              _i32, _vp, _vp],
    "eg_project_hits": [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp],
    "eg_project_visibility": [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp],
    "eg_knn": [_vp, _i32, _i32, C.POINTER(_f), _f, C.POINTER(_i32), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eg_knn_small": [_vp, _i32, _i32, _vp, _vp, _vp],
    "eg_knn_auto": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "eg_direction_loss": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "eg_ratio_loss": [_vp, _i32, _vp, _vp, _vp],
    "eg_regulariser_step": [_i32] + [_vp] * 7 + [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _f, _f, _vp, AdamHyper, _vp],
    "eg_regulariser_step_fixed": [_i32] + [_vp] * 7 + [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _f, _f, _vp, AdamHyper, _vp, _vp],
    "eg_train_step": [C.POINTER(StepArgs), _vp],
    "eg_train_steps": [C.POINTER(StepArgs), _i32, C.POINTER(_i32), C.POINTER(_vp), _vp, _vp, _vp, _vp],
    # (S scenes: arrays of S pointers -- args, views [K], weight-map tables [K], viewmats, Ks, gts, streams -- + n_threads)
    "eg_train_steps_multi": [_i32, C.POINTER(C.POINTER(StepArgs)), _i32, C.POINTER(C.POINTER(_i32)), C.POINTER(C.POINTER(_vp)),
                             C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i32],
    "eg_operator_fwd": [C.POINTER(OperatorArgs), _vp],
    "eg_operator_bwd": [C.POINTER(OperatorArgs), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eg_train_steps_dp": [C.POINTER(StepArgs), C.POINTER(AdamHyper), _vp, _i32, C.POINTER(_i32), C.POINTER(_vp), _vp, _vp,
                          _vp, _i32, _vp],
    "eg_dp_all_reduce": [_vp, _i64, _vp],
    "eg_project_fwd_cams": [_vp] * 6 + [_i32, _i32, _i32, _i32, _f, _f, _f, _f, _u32] + [_vp] * 8 + [_vp],
    "eg_project_bwd_cams": [_vp] * 6 + [_i32, _i32, _i32, _i32, _f, _u32] + [_vp] * 7 + [_vp],
    "eg_tile_offsets_cams": [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp],
    "eg_tile_emit_sort_cams": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp),
                               C.POINTER(_vp), C.POINTER(_i32), _vp],
    "eg_composite_fwd_cams": [_i32, _vp, _i32, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp), _i32, _i32, _vp, _vp, _vp,
                              C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp), _vp, _vp],
    "eg_composite_bwd_footprint_cams": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "eg_train_step_batched": [C.POINTER(StepArgs), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp],
}
EXPORTS = sorted(list(_SIGS) + ["eg_last_error_string", "eg_version", "eg_device_count",
                                 "eg_composite_workspace_bytes", "eg_composite_workspace_ctl_bytes",
                                 "eg_batched_workspace_stride", "eg_knn_auto_dims", "eg_timing_begin", "eg_timing_end",
                                 "eg_timing_stage_count", "eg_timing_stage_name", "eg_debug_fwd_profile",
                                 "eg_dp_unique_id", "eg_dp_init", "eg_dp_world", "eg_dp_shutdown", "eg_dp_host_profile", "eg_roctx_enable",
                                 "eg_dp_comm_count", "eg_dp_force_all_reduce", "eg_dp_grad_all_reduces",
                                 "eg_dp_comm_timing_begin", "eg_dp_comm_timing_end", "eg_record_xcd_shift",
                                 "eg_backward_is_fused"])

_lib: Optional[C.CDLL] = None


def load(require_device: bool = True) -> C.CDLL:
    """Loads the library (once).  Raises RuntimeError when it is missing -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m edgegaussians_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback in the product path.")
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.eg_last_error_string.restype = C.c_char_p
        lib.eg_last_error_string.argtypes = []
        lib.eg_version.restype = C.c_int
        lib.eg_device_count.restype = C.c_int
        lib.eg_timing_begin.argtypes = [_i32]
        lib.eg_timing_end.argtypes = [C.POINTER(C.c_float), C.POINTER(_i32)]
        lib.eg_timing_stage_name.argtypes = [_i32]
        lib.eg_timing_stage_name.restype = C.c_char_p
        lib.eg_composite_workspace_bytes.restype = _i64
        lib.eg_composite_workspace_bytes.argtypes = [_i64, _i64]
        lib.eg_composite_workspace_ctl_bytes.restype = _i64
        lib.eg_composite_workspace_ctl_bytes.argtypes = [_i64, _i64]
        lib.eg_batched_workspace_stride.restype = _i64
        lib.eg_batched_workspace_stride.argtypes = [_i64, _i64]
        lib.eg_knn_auto_dims.restype = _i32
        lib.eg_knn_auto_dims.argtypes = [_i32, _i32]
        lib.eg_dp_unique_id.argtypes = [C.c_char_p, _vp]
        lib.eg_dp_init.argtypes = [C.c_char_p, _vp, _i32, _i32]
        lib.eg_dp_world.argtypes = []
        lib.eg_dp_shutdown.argtypes = []
        lib.eg_roctx_enable.argtypes = [_i32]
        lib.eg_record_xcd_shift.argtypes = [_i32]
        lib.eg_backward_is_fused.argtypes = [_i32, _i32]
        lib.eg_dp_comm_count.argtypes = []
        lib.eg_dp_force_all_reduce.argtypes = [_i32]
        lib.eg_dp_grad_all_reduces.argtypes = [C.POINTER(_i64)]
        lib.eg_dp_grad_all_reduces.restype = _i64
        lib.eg_dp_comm_timing_begin.argtypes = [_i32]
        lib.eg_dp_comm_timing_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_i32)]
        lib.eg_dp_host_profile.argtypes = [C.POINTER(C.c_double)]
        lib.eg_dp_host_profile.restype = _i64
        lib.eg_debug_fwd_profile.restype = _i64
        lib.eg_debug_fwd_profile.argtypes = [_vp, _i64]
        _lib = lib
    if require_device and not torch.cuda.is_available():
        raise RuntimeError("edgegaussians_amd needs a gfx950 GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback in the product path")
    return _lib


def composite_workspace(max_items: int, n_tiles: int, device) -> torch.Tensor:
    """Scratch of the slice-parallel forward for (max_items, n_tiles), control words zeroed (include/edgegs.h)."""
    lib = load()
    # (all of it zeroed: the hand-over granules of the wave-autonomous forward carry a tag, and tag 0 means "never written")
    return torch.zeros(lib.eg_composite_workspace_bytes(max_items, n_tiles), dtype=torch.uint8, device=device)


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.eg_last_error_string().decode()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libedgegs needs contiguous device tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The current torch stream's raw handle (what every entry point takes as its last argument)."""
    if _raw_stream is not None:  # ~1 us instead of ~9 us for the Stream-object round trip: it is paid per step
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
