"""Build libcurobo_hip.so (the C-ABI kernel backend) with hipcc for gfx950.

The shared object is built in-tree (``curobo_amd/lib/libcurobo_hip.so``) so that it travels with
the source snapshot to the GPU box; hipcc cross-compiles for gfx950 without a GPU.
Usage: ``python -m curobo_amd.build [--force]``.
"""

from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from typing import List

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcurobo_hip.so")
OBJ_DIR = os.path.join(_PKG, "build")
ARCH = "gfx950"

HEADERS = ["common.hpp", "cost_device.hpp", "scene_device.hpp", "fk_device.hpp", "self_device.hpp",
           "bspline_device.hpp", "dynamics_device.hpp", "mesh_device.hpp", "fused_shapes.hpp"]

SOURCES = [
    "runtime.cpp",
    "kinematics.hip",
    "self_collision.hip",
    "scene_collision.hip",
    "trajectory.hip",
    "optimization.hip",
    "cost.hip", "rollout_fused.hip", "dynamics.hip", "linalg.hip", "mppi.hip", "seed_ik.hip", "mesh_bake.hip", "mesh_bvh.hip",
]


def fused_shape_ids() -> List[int]:
    """Compile-time shapes of the fused rollout launch (csrc/fused_shapes.hpp: CUROBO_FUSED_NUM_SHAPES)."""
    import re

    m = re.search(r"#define\s+CUROBO_FUSED_NUM_SHAPES\s+(\d+)", open(os.path.join(CSRC, "fused_shapes.hpp")).read())
    return list(range(1, int(m.group(1)) + 1)) if m else []


def compile_units() -> List[tuple]:
    """(source, object stem, extra flags): every source once, plus rollout_fused.hip once more per compile-time shape
    (-DCUROBO_FUSED_SHAPE_TU=k holds only that shape's instantiations and its launcher: the shapes compile in parallel)."""
    units = [(s, s.rsplit(".", 1)[0], []) for s in SOURCES]
    units += [("rollout_fused.hip", f"rollout_fused_shape{k}", [f"-DCUROBO_FUSED_SHAPE_TU={k}"]) for k in fused_shape_ids()]
    return units


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP backend cannot be built on this machine")


def _flags() -> List[str]:
    return [
        f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
        "-ffp-contract=fast", "-fno-fast-math", "-Wall", "-Wno-unused-function",
        # fp32 div/sqrt as single hardware ops (<= 2.5 ulp) instead of the IEEE fix-up sequences;
        # the reference kernels are built the same way (--prec-div=false --prec-sqrt=false,
        # cuda_core_backend/kernel_cache.py:208-218)
        "-fno-hip-fp32-correctly-rounded-divide-sqrt",
        f"-I{INCLUDE}", f"-I{CSRC}",
        # diagnostic builds only (e.g. CUROBO_HIP_EXTRA_FLAGS=-DCUROBO_FUSED_STAMP_TERMS with force=True)
        *os.environ.get("CUROBO_HIP_EXTRA_FLAGS", "").split(),
    ]


# The SLP vectoriser packs independent fp32 products into v_pk_* pairs; on gfx950 a packed fp32 instruction issues at
# half rate and every pair costs register shuffles (v_mov / v_pk_mov): ~110 instructions per link become ~150 on the
# serial chain of kinematics.hip, and rollout_fused.hip's main instantiation needs 125 instead of 95 VGPRs (its
# SWEEP x voxel instantiations spill 54-60 registers with it, 2-4 without).  Off for every source.
# self_collision.hip: the matrix-core narrow phase compares the tile results on the vector ALU, so they must land in VGPRs -- the
# AGPR form costs a v_accvgpr_write per accumulator input and a v_accvgpr_read per output, twelve moves per 16 x 16 tile
PER_SOURCE_FLAGS: dict = {"self_collision.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
NO_SLP = ["-fno-slp-vectorize"]


def _included(path: str, seen: set) -> None:
    """the package headers a source reaches through its #include "..." lines (transitively)"""
    import re

    try:
        text = open(path).read()
    except OSError:
        return
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for base in (CSRC, INCLUDE):
            cand = os.path.join(base, name)
            if os.path.exists(cand) and cand not in seen:
                seen.add(cand)
                _included(cand, seen)


def _deps(src: str) -> List[str]:
    """the source, the C header and the device headers it actually includes (a header edit rebuilds its users, not the library)"""
    seen: set = set()
    _included(src, seen)
    return [src, os.path.join(INCLUDE, "curobo_hip.h"), *sorted(seen)]


def _compile(unit, force: bool) -> str:
    src_name, stem, extra = unit if isinstance(unit, tuple) else (unit, unit.rsplit(".", 1)[0], [])
    src = os.path.join(CSRC, src_name)
    obj = os.path.join(OBJ_DIR, stem + ".o")
    if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in _deps(src)):
        return obj
    cmd = [hipcc_path(), *_flags(), *NO_SLP, *PER_SOURCE_FLAGS.get(src_name, []), *extra, "-x", "hip", "-c", src, "-o", obj]
    subprocess.check_call(cmd)
    return obj


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(INCLUDE, "curobo_hip.h")] + [
        os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    units = compile_units()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:
        objs = list(ex.map(lambda u: _compile(u, force), units))
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB_PATH)
    shutil.copyfile(os.path.join(INCLUDE, "curobo_hip.h"), os.path.join(LIB_DIR, "curobo_hip.h"))  # travels with the library
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
