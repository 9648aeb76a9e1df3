"""Compile-time shapes of the fused rollout launch compiled at RUN TIME for the robot / horizon at hand.

The library ships a handful of pre-built shapes (``csrc/fused_shapes.hpp``: Franka and UR10e at the BASELINE horizons); any
other robot, horizon or knot count runs the generic kernel, which is ~20 % slower (every dimension a run-time value).  This
module builds the missing shape on demand -- ``hipcc`` on ``csrc/rollout_fused.hip`` with the shape and its kernel list on the
command line (5-10 s, cached on disk by a hash of the sources, the shape, the flags and the compiler), the object loaded and its
launcher handed to ``curobo_hip_rollout_fused_register_shape`` -- which is what the reference does for every kernel and every
robot with NVRTC (``curobo/_src/curobolib/backends/cuda_core_backend/kernel_cache.py:89-119,161-235``: SHA-256 of sources + name
+ flags + arch, compiled once, cached).  Opt-in: ``CollisionRolloutCfg(jit_shape=True)`` / ``TrajOptRolloutCfg(jit_shape=True)``
or ``CUROBO_HIP_JIT_SHAPES=1``; a failed build (no compiler on the machine) leaves the launch on the generic kernel.
"""

from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import threading
from typing import Dict, Iterable, Optional, Tuple

from .._lib import check, load

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_PKG), "include")
_loaded: Dict[str, C.CDLL] = {}
_lock = threading.Lock()

#: (bspline degree, sweep steps, obstacle kinds, with trajopt terms) instantiations a run-time shape holds by default
DEFAULT_KERNELS = ((3, 3, 1, False), (3, 3, 1, True), (3, 3, 2, False), (3, 3, 3, False))


def cache_dir() -> str:
    d = os.environ.get("CUROBO_HIP_JIT_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "curobo_amd", "fused_shapes")
    os.makedirs(d, exist_ok=True)
    return d


def enabled_by_env() -> bool:
    return os.environ.get("CUROBO_HIP_JIT_SHAPES", "") not in ("", "0")


def _flags():
    from ..build import NO_SLP, _flags as build_flags

    return [f for f in build_flags() if not f.startswith("-I")] + [f"-I{_INCLUDE}", f"-I{_CSRC}", *NO_SLP]


def _sources_digest() -> str:
    h = hashlib.sha256()
    names = ["rollout_fused.hip"] + sorted(n for n in os.listdir(_CSRC) if n.endswith(".hpp"))
    for n in names:
        with open(os.path.join(_CSRC, n), "rb") as fh:
            h.update(n.encode())
            h.update(fh.read())
    hdr = os.path.join(_INCLUDE, "curobo_hip.h")
    if not os.path.exists(hdr):
        hdr = os.path.join(_PKG, "lib", "curobo_hip.h")
    with open(hdr, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def shape_spec(padded_horizon: int, n_knots: int, dof: int, num_links: int, num_spheres: int, num_collision_pairs: int,
               link_chain_len: int, self_lane_len: int, threads: int, max_cuboids: int = -1, max_voxel_grids: int = -1,
               plain: bool = False) -> str:
    """the ``FusedShape<...>`` instantiation of these dimensions (``csrc/fused_shapes.hpp``)"""
    return (f"FusedShape<{int(padded_horizon)}, {int(n_knots)}, {int(dof)}, {int(num_links)}, {int(num_spheres)}, "
            f"{int(num_collision_pairs)}, {int(link_chain_len)}, {int(self_lane_len) & 0xffff}, {(int(self_lane_len) >> 16) & 0xffff}, "
            f"{int(threads)}, {int(max_cuboids)}, {int(max_voxel_grids)}, {1 if plain else 0}>")


def compile_shape(spec: str, kernels: Iterable[Tuple[int, int, int, bool]] = DEFAULT_KERNELS, verbose: bool = False) -> str:
    """Build (or find in the cache) the shared object of one shape; returns its path.  Raises when there is no compiler."""
    from ..build import ARCH, hipcc_path

    klist = " ".join(f"K({int(d)}, {int(s)}, {int(k)}, {'true' if t else 'false'})" for d, s, k, t in kernels)
    flags = _flags()
    cc = hipcc_path()
    key = hashlib.sha256("\n".join([_sources_digest(), spec, klist, " ".join(flags), cc, ARCH]).encode()).hexdigest()[:32]
    out = os.path.join(cache_dir(), f"fused_shape_{key}.so")
    if os.path.exists(out):
        return out
    tmp = f"{out}.{os.getpid()}.tmp"
    cmd = [cc, *flags, "-shared", "-DCUROBO_FUSED_SHAPE_TU=99", f"-DCUROBO_FUSED_JIT_SHAPE={spec}", f"-DCUROBO_FUSED_JIT_KERNELS(K)={klist}",
           "-x", "hip", os.path.join(_CSRC, "rollout_fused.hip"), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        try:
            os.remove(tmp)
        except OSError:
            pass
        raise RuntimeError(f"hipcc failed on the run-time shape {spec}:\n{p.stderr[-2000:]}")
    os.replace(tmp, out)  # (atomic: concurrent processes -- one per GPU -- may build the same shape)
    return out


def register_shape_object(path: str) -> None:
    """load a shape object and register its launcher with the library (idempotent)"""
    with _lock:
        if path in _loaded:
            return
        so = C.CDLL(path)
        so.curobo_fused_jit_args_bytes.restype = C.c_int
        fn = C.cast(so.curobo_fused_jit_launch, C.c_void_p)
        check(load().curobo_hip_rollout_fused_register_shape(fn, int(so.curobo_fused_jit_args_bytes())))
        _loaded[path] = so  # (keeps the object -- its kernels -- alive for the life of the process)


def ensure_shape(padded_horizon: int, n_knots: int, dof: int, num_links: int, num_spheres: int, num_collision_pairs: int,
                 link_chain_len: int, self_lane_len: int, num_obstacles: int, with_trajopt_terms: bool = False,
                 kernels: Optional[Iterable[Tuple[int, int, int, bool]]] = None, verbose: bool = False) -> bool:
    """Make sure launches of these dimensions have a compile-time shape: nothing to do when the library (or an earlier call)
    already holds one for them, else compile + register the PLAIN form (an optimiser iteration: collision-only, or with
    ``with_trajopt_terms`` the full trajectory-optimisation cost set) and the any-form shape.
    ``self_lane_len`` = what ``curobo_hip_self_lane_lists_host`` returned for the robot (0: no lane lists -> no shapes: the
    shapes are built for the lane form of the pair pass).  Returns whether a shape now serves these dimensions; a build
    failure is reported once and leaves the generic kernel in place."""
    from .rollout import fused_shape_id

    if not self_lane_len or num_collision_pairs <= 0:
        return False
    lib = load()
    have = lambda plain: fused_shape_id(padded_horizon, n_knots, dof, num_links, num_spheres, num_collision_pairs, link_chain_len,  # noqa: E731
                                        self_lane_len, max(num_obstacles, 1), 0, with_trajopt_terms=with_trajopt_terms, plain_launch=plain)
    ok = True
    for plain in (True, False):
        if have(plain) != 0:
            continue
        threads = int(lib.curobo_hip_rollout_fused_threads(int(padded_horizon), int(dof), int(num_links), int(num_spheres), int(num_collision_pairs),
                                                           int(link_chain_len), int(self_lane_len), int(num_obstacles), 1 if with_trajopt_terms else 0))
        spec = shape_spec(padded_horizon, n_knots, dof, num_links, num_spheres, num_collision_pairs, link_chain_len, self_lane_len, threads,
                          -1, -1, plain)
        # a plain shape holds the instantiations of ITS launch form: collision-only rollouts or the full trajopt cost set
        ks = tuple(kernels) if kernels is not None else tuple(k for k in DEFAULT_KERNELS if not plain or k[3] == bool(with_trajopt_terms))
        try:
            register_shape_object(compile_shape(spec, ks, verbose=verbose))
        except Exception as e:  # noqa: BLE001  (no compiler, a full disk ...: the generic kernel keeps serving the launch)
            global _warned
            if not _warned:
                import warnings

                warnings.warn(f"curobo_amd: run-time shape compilation failed, the generic fused kernel stays in use ({type(e).__name__}: "
                              f"{str(e)[:300]})")
                _warned = True
            ok = False
    return ok


_warned = False
