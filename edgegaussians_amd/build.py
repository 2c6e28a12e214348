"""In-tree build of libedgegs.so (hand-written HIP for gfx950) with plain hipcc.

    python -m edgegaussians_amd.build            # build if stale
    python -m edgegaussians_amd.build --force

hipcc cross-compiles gfx950 without a GPU; the .so lands next to this file so that it travels
with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libedgegs.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["project.hip", "binning.hip", "composite.hip", "composite_wave.hip", "densify.hip", "knn.hip", "step.hip", "dp.hip", "operator.hip", "backward_fused.hip", "cams.hip"]
# -fno-slp-vectorize: the SLP pass pairs scalar fp32 operations into v_pk_*_f32, which gfx950's vector pipe issues at
# exactly the cost of the two plain instructions (tools/microbench/issue_rates.hip: 5.6 vs 2 x 2.8 cycles) -- and the
# pairs need their operands in adjacent registers: v_mov shuffles and s_nop hazards on top.  Measured on the whole step
# (profiles/r03_slp_ab.txt): footprint backward -12 % on top of its own rewrite, projection backward -4 %.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function"]
# EG_DEV_SWITCHES=1 at BUILD time compiles the A/B environment switches of the kernels' launchers in (NOTES_r04.md);
# a release build reads no environment variable on the product path except EG_FWD_PROF (the profiling instantiation)
if os.environ.get("EG_DEV_SWITCHES"):
    FLAGS.append("-DEG_DEV_SWITCHES")
FLAGS += os.environ.get("EG_EXTRA_HIPCC_FLAGS", "").split()  # (development: compile-time A/B legs)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libedgegs.so cannot be built")


STAMP = os.path.join(OBJ, "flags.txt")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(STAMP) or open(STAMP).read() != " ".join(FLAGS):
        return True  # (built with other flags, e.g. a development build with EG_DEV_SWITCHES)
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + ["common.h", "composite.h", "project_dev.h", "footprint_dev.h"]]
    deps.append(os.path.join(HERE, "..", "include", "edgegs.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(" ".join(FLAGS))
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
