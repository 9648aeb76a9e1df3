"""ctypes binding of ``libcurobo_hip.so`` (the C ABI declared in ``include/curobo_hip.h``).

The product path fails loudly when the HIP library is missing: there is no CPU or eager-torch
fallback behind these calls.
"""

from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libcurobo_hip.so")
# the public header: the repository's include/ directory, else the copy the build puts next to the library (a build
# output, untracked: it makes a copied / installed package self-contained).  load() refuses a library whose header copy
# differs from include/curobo_hip.h: the argtypes below are parsed from the header and must describe the loaded .so.
_HEADER_CANDIDATES = (os.path.join(os.path.dirname(_PKG), "include", "curobo_hip.h"), os.path.join(_PKG, "lib", "curobo_hip.h"))
HEADER_PATH = next((p for p in _HEADER_CANDIDATES if os.path.exists(p)), _HEADER_CANDIDATES[0])


def _header_is_stale() -> bool:
    """include/curobo_hip.h was edited after the library (and its header copy) were built"""
    a, b = _HEADER_CANDIDATES
    if not (os.path.exists(a) and os.path.exists(b)):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        return fa.read() != fb.read()

_lib: Optional[C.CDLL] = None


class CuroboHipError(RuntimeError):
    """Launch failure reported by the HIP backend (reference: log_and_raise after
    cudaGetLastError, cuda_core_backend/launch_helper.py:17-19)."""


class Scene(C.Structure):
    """``curobo_hip_scene`` (see include/curobo_hip.h)."""

    _fields_ = [
        ("cuboid_dims", C.c_void_p), ("cuboid_inv_pose", C.c_void_p), ("cuboid_enable", C.c_void_p),
        ("cuboid_count", C.c_void_p), ("max_cuboids", C.c_int32),
        ("voxel_params", C.c_void_p), ("voxel_inv_pose", C.c_void_p), ("voxel_enable", C.c_void_p),
        ("voxel_count", C.c_void_p), ("voxel_features", C.c_void_p), ("max_voxel_grids", C.c_int32),
        ("voxel_n_voxels", C.c_int32), ("voxel_max_distance", C.c_float),
        ("voxel_coarse_min", C.c_void_p), ("voxel_coarse_block", C.c_int32), ("voxel_coarse_dilate", C.c_int32),
        ("voxel_n_coarse", C.c_int32), ("cuboid_has_primitives", C.c_int32),
    ]


class TrajOptTerms(C.Structure):
    """``curobo_hip_trajopt_terms`` (see include/curobo_hip.h): optional tool-pose / c-space STATE
    terms of the fused trajopt rollout.  Field order and types mirror the C struct."""

    _P, _I = C.c_void_p, C.c_int32
    _fields_ = [
        ("out_pose_distance", _P), ("out_position_distance", _P), ("out_rotation_distance", _P), ("out_goalset_idx", _P),
        ("goal_position", _P), ("goal_quat", _P), ("idxs_goal", _P), ("position_orientation_weight", _P),
        ("terminal_pose_axes_weight_factor", _P), ("non_terminal_pose_axes_weight_factor", _P),
        ("terminal_pose_convergence_tolerance", _P), ("non_terminal_pose_convergence_tolerance", _P),
        ("project_distance_to_goal", _P), ("tool_frame_map", _P),
        ("n_tool_frames", _I), ("num_goalset", _I), ("rotation_method", _I),
        ("out_cspace_cost", _P), ("state_dt", _P), ("target_joint_position", _P), ("idxs_target_joint_position", _P),
        ("p_b", _P), ("v_b", _P), ("a_b", _P), ("j_b", _P), ("effort_b", _P),
        ("cspace_weight", _P), ("cspace_activation_distance", _P), ("squared_l2_regularization_weights", _P),
        ("cspace_target_weight", _P), ("cspace_non_terminal_weight_factor", _P), ("cspace_target_dof_weight", _P),
        ("retime_weights", _I), ("retime_regularization_weights", _I),
        ("link_masses_com", _P), ("link_inertias", _P), ("gravity", _P), ("level_links", _P), ("use_torque_limits", _I),
    ]


def declared_symbols(header: str = HEADER_PATH) -> List[str]:
    """Every function name the public header declares."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(curobo_hip_[a-z0-9_]+)\s*\(", text)))


def _ctype_of(tok: str):
    tok = tok.strip()
    if "*" in tok or tok.startswith("curobo_hip_stream_t"):
        return C.c_void_p
    if tok.startswith("float"):
        return C.c_float
    if tok.startswith("size_t"):
        return C.c_size_t
    if tok.startswith("int64_t"):
        return C.c_int64
    if tok.startswith("int") or tok.startswith("int32_t"):
        return C.c_int
    raise ValueError(f"unhandled C type in header: {tok!r}")


def _signatures(header: str = HEADER_PATH) -> Dict[str, list]:
    """Parse argument types from the header so Python and C cannot drift apart."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\bint\s+(curobo_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if args in ("", "void"):
            sigs[name] = []
            continue
        sigs[name] = [_ctype_of(a) for a in args.split(",")]
    return sigs


def library_available() -> bool:
    return os.path.exists(LIB_PATH)


def load() -> C.CDLL:
    """Load the shared library (after ``import torch`` so torch's HIP runtime is reused)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m curobo_amd.build` "
            "(hipcc --offload-arch=gfx950). curobo_amd has no CPU fallback."
        )
    if _header_is_stale():
        raise ImportError(
            f"{_HEADER_CANDIDATES[0]} differs from the header {LIB_PATH} was built with ({_HEADER_CANDIDATES[1]}): "
            "rebuild with `python -m curobo_amd.build` so that the ctypes argument types match the library")
    import torch  # noqa: F401  (loads libamdhip64 first; same SONAME is then shared)

    lib = C.CDLL(LIB_PATH)
    sigs = _signatures()
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    # every declared entry point must have been understood by the header parser (a formatting the
    # regex misses would otherwise leave a function without argtypes: silent int truncation of pointers)
    special = {"curobo_hip_last_error", "curobo_hip_set_debug_sync"}
    untyped = [n for n in declared_symbols() if n not in sigs and n not in special]
    if untyped:
        raise ImportError(f"include/curobo_hip.h declares entry points the ctypes binding could not parse: {untyped}")
    lib.curobo_hip_last_error.restype = C.c_char_p
    lib.curobo_hip_last_error.argtypes = []
    lib.curobo_hip_set_debug_sync.restype = None
    lib.curobo_hip_set_debug_sync.argtypes = [C.c_int]
    _lib = lib
    return lib


def check(status: int) -> None:
    """Convert a C status into the reference's exception types."""
    if status == 0:
        return
    msg = load().curobo_hip_last_error().decode()
    if status == 1:
        raise ValueError(msg)
    raise CuroboHipError(msg)


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(t) -> int:
    """Raw hipStream_t of torch's current stream on the tensor's device.

    Mandatory for correctness under stream-per-cost execution and graph capture (reference
    cuda_core_backend/kinematics.py:50-51)."""
    import torch

    if t.device.index != torch.cuda.current_device():
        # the launch (and hipFuncSetAttribute) would go to the current device's context
        raise ValueError(f"tensor on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                         "wrap the call in `with torch.cuda.device(tensor.device):`")
    return torch.cuda.current_stream(t.device).cuda_stream
