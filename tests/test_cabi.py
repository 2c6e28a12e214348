"""The C-ABI library loads and exports every symbol include/edgegs.h declares (no compute calls:
there is no GPU here), and the product path fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "edgegs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eg_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from edgegaussians_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25
    h = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing


def test_python_binding_covers_the_header(lib):
    assert sorted(lib.EXPORTS) == _declared()


def test_library_reports_no_device_and_product_refuses(lib):
    import torch
    h = lib.load(require_device=False)
    assert h.eg_version() >= 100
    assert h.eg_last_error_string() is not None
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert h.eg_device_count() <= 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load(require_device=True)
    from edgegaussians_amd import rasterization
    with pytest.raises((RuntimeError, ValueError)):
        rasterization(torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 3), torch.zeros(4), torch.ones(4, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 32, 32, packed=False)


def test_argument_validation_without_launching(lib):
    h = lib.load(require_device=False)
    # null pointers / bad sizes are rejected before any HIP call
    assert h.eg_tile_offsets(None, 0, 0, None, None, None, None) == -1
    assert b"eg_tile_offsets" in h.eg_last_error_string()
    ctl = h.eg_composite_workspace_ctl_bytes(10, 4)
    # per-tile tickets, item flags + tags, 4 control words, 64 exit-ticket lines; then the words of the wave-autonomous
    # forward (per-quadrant tickets and tags, 64 partial loss sums, the dead-slice hints), rounded up to 16 bytes: the
    # hand-over granules behind them are single 8-byte words
    words = (4 + 2 * 10 + 4 + 64 * 16) + (4 * 4 + 4 * 10 + 64 + 8 * 4)
    assert ctl == (words + 3) // 4 * 16 and ctl % 16 == 0
    assert h.eg_composite_workspace_ctl_bytes(10, 117) % 16 == 0  # (an odd tile count once mis-aligned the granules)
    # ... the aggregate granules, the general path's stop hand-over / re-walk list / quadrant verdicts, and (8-byte
    # aligned) the inclusive granules of the anchor slices: one block per 8 items + 2
    assert h.eg_composite_workspace_bytes(10, 4) == ctl + 10 * 256 * 8 + 4 * 256 * 12 + 10 * 8 + 10 * 128 + 8 + (10 // 8 + 2) * 256 * 8
    assert h.eg_timing_stage_count() == 7 and h.eg_timing_stage_name(5) == b"footprint_bwd"
    # the footprint backward addresses the record image with 32-bit byte offsets: an image it could not address is
    # refused, not truncated (46341^2 pixels x 12 bytes >= 2^31)
    assert h.eg_composite_bwd_footprint(None, 8, 46341, 46341, None, None, None) == -1
    assert b"2^31" in h.eg_last_error_string()
    # the single-tensor Adam behind edgegaussians_amd.optim.Adam: sizes and the step count are checked first
    assert h.eg_adam_tensor(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0, 0, None) == -1
    assert h.eg_adam_tensor(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 1, 0, None) == -1  # null pointers
    assert h.eg_adam_tensor(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 1, 0, None) == 0    # nothing to do


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "edgegaussians_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dirpath, f)).read(), re.M):
                bad.append(f)
    for f in os.listdir(os.path.join(ROOT, "gsplat")):
        if f.endswith(".py") and "oracle" in open(os.path.join(ROOT, "gsplat", f)).read():
            bad.append(f)
    assert not bad, bad
