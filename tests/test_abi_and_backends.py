"""The C-ABI library loads on a CPU-only machine and exports every symbol the header declares;
the Python backend modules expose the reference's function names with the reference's
positional argument order (checked against the reference sources when they are present)."""

import ast
import ctypes
import inspect
import os

import pytest

from curobo_amd import _lib

REF_BACKEND = "/root/reference/curobo/_src/curobolib/backends/cuda_core_backend"


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"libcurobo_hip.so does not export {n}"
    assert lib.curobo_hip_abi_version() == 7
    assert isinstance(lib.curobo_hip_last_error(), bytes)


def test_header_signatures_are_plain_c():
    sigs = _lib._signatures()
    assert set(sigs) >= {"curobo_hip_launch_kinematics_forward_spheres", "curobo_hip_self_collision_distance",
                         "curobo_hip_launch_lbfgs_step", "curobo_hip_launch_line_search"}
    for name, args in sigs.items():
        for a in args:
            # plain pointers, ints, floats and sizes (size_t / int64_t byte counts): nothing a C FFI cannot bind
            assert a in (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64), (name, a)


def test_invalid_arguments_are_rejected_without_a_gpu():
    """argument validation happens before any launch (reference raises ValueError/RuntimeError)"""
    lib = _lib.load()
    rc = lib.curobo_hip_launch_lbfgs_step(None, None, None, None, None, None, None, None, 0.01, 4, 32, 84, 1, 1, None)
    assert rc == 1 and b"History_m greater than 31" in lib.curobo_hip_last_error()
    with pytest.raises(ValueError, match="History_m"):
        _lib.check(rc)
    rc = lib.curobo_hip_launch_kinematics_backward(*([None] * 23), 1, 10, 1, 7, 65, 13, 1, 88, 0, 1, None)
    assert rc == 1 and b"compute_jacobian_grad" in lib.curobo_hip_last_error()


EXPECTED = {
    "kinematics": ["launch_kinematics_forward", "launch_kinematics_forward_spheres",
                   "launch_kinematics_forward_spheres_jacobian", "launch_kinematics_backward"],
    "geometry": ["self_collision_distance"],
    "trajectory": ["launch_bspline_interpolation_forward_kernel", "launch_bspline_interpolation_backward_kernel",
                   "launch_bspline_interpolation_single_dt_kernel", "launch_differentiation_position_forward_kernel",
                   "launch_differentiation_position_backward_kernel", "launch_integration_acceleration_kernel"],
    "optimization": ["launch_line_search", "launch_lbfgs_step"],
    "dynamics": ["launch_rnea_forward", "launch_rnea_backward"],
}
# HIP-side extensions (no reference backend module of that name): exported and callable
EXTENSIONS = {
    "cost": ["tool_pose_distance", "cspace_position_cost", "cspace_state_cost", "cspace_l2_distance", "rollout_point_aggregate"],
    "linalg": ["levenberg_marquardt_step", "seed_ik_update_state"],
    "rollout": ["rollout_trajectory_fused", "rollout_trajopt_fused", "rollout_ik_fused", "make_trajopt_terms", "DispatchOrder"],
    "optimization": ["launch_lbfgs_iteration_tail", "prepare_search_points", "mppi_update_distribution"],
}


@pytest.mark.parametrize("module", sorted(EXPECTED))
def test_backend_modules_export_reference_names(module):
    import importlib

    mod = importlib.import_module(f"curobo_amd.backends.{module}")
    for fn in EXPECTED[module]:
        assert callable(getattr(mod, fn))


@pytest.mark.parametrize("module", sorted(EXTENSIONS))
def test_extension_backend_modules(module):
    import importlib

    mod = importlib.import_module(f"curobo_amd.backends.{module}")
    for fn in EXTENSIONS[module]:
        assert callable(getattr(mod, fn)), f"{module}.{fn}"


def _ref_signature(module, fn):
    tree = ast.parse(open(os.path.join(REF_BACKEND, f"{module}.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == fn:
            return [a.arg for a in node.args.args]
    raise KeyError(fn)


@pytest.mark.skipif(not os.path.isdir(REF_BACKEND), reason="reference checkout not present")
@pytest.mark.parametrize("module,fn", [(m, f) for m in sorted(EXPECTED) for f in EXPECTED[m]])
def test_positional_arguments_match_reference(module, fn):
    import importlib

    ours = list(inspect.signature(getattr(importlib.import_module(f"curobo_amd.backends.{module}"), fn)).parameters)
    ref = _ref_signature(module, fn)
    assert ours[: len(ref)] == ref, f"{module}.{fn}: positional arguments differ from the reference backend"
    extra = ours[len(ref):]
    assert all(inspect.signature(getattr(importlib.import_module(f"curobo_amd.backends.{module}"), fn)).parameters[e].default
               is not inspect.Parameter.empty for e in extra), "extensions must be optional keywords"


def test_get_backend_shape():
    from curobo_amd.backends import get_backend

    be = get_backend()
    assert {"kinematics", "optimization", "trajectory", "geometry"} <= set(be)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under curobo_amd/ may import it (statically checked on
    every module's AST, so a lazily imported fallback would be caught as well)."""
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    bad = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            for node in ast.walk(ast.parse(open(path).read(), path)):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.level == 0:
                    names = [node.module or ""]
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    bad.append(os.path.relpath(path, root))
    assert not bad, f"product modules import the oracle: {bad}"


def test_missing_library_fails_loudly(monkeypatch):
    """no silent CPU / eager fallback: without the built HIP library every entry point raises"""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(os.path.dirname(_lib.LIB_PATH), "no_such_library.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    from curobo_amd.backends import kinematics

    n_args = len([p for p in inspect.signature(kinematics.launch_kinematics_forward).parameters.values()
                  if p.default is inspect.Parameter.empty])
    with pytest.raises(ImportError, match="no CPU fallback"):
        kinematics.launch_kinematics_forward(*([None] * n_args))


def test_rnea_walk_order_is_depth_first_and_survives_address_reuse():
    """backends/dynamics._walk_order: parents before children, a child right after its parent wherever the tree allows
    (what the kernels' register forwarding relies on), cached per link table, and a cache entry whose tensor died is
    not trusted (the allocator hands the address to the next robot's table of the same size)."""
    import gc
    import weakref

    import torch

    from curobo_amd.backends import dynamics as D

    #        0
    #      /   \
    #     1     4
    #    / \     \
    #   2   3     5
    parent = torch.tensor([-1, 0, 1, 1, 0, 4], dtype=torch.int16)
    level = torch.tensor([0, 1, 4, 2, 3, 5], dtype=torch.int16)  # breadth-first, what the loader produces
    order = D._walk_order(parent, level).tolist()
    assert sorted(order) == list(range(6))
    pos = {k: i for i, k in enumerate(order)}
    for k, p in enumerate(parent.tolist()):
        assert p < 0 or pos[p] < pos[k]
    assert order == [0, 1, 2, 3, 4, 5]
    assert D._walk_order(parent, level) is D._walk_order(parent, level)  # cached
    # the same key with a dead owner: recomputed from the tensor that is passed in
    key = (parent.data_ptr(), int(parent.numel()), str(parent.device))
    ghost = torch.zeros(1)
    D._dfs_orders[key] = (weakref.ref(ghost), torch.tensor([5, 4, 3, 2, 1, 0], dtype=torch.int16))
    del ghost
    gc.collect()
    assert D._walk_order(parent, level).tolist() == [0, 1, 2, 3, 4, 5]
    # not a forest (a cycle): the caller's level order is kept
    cyc = torch.tensor([1, 0, 1], dtype=torch.int16)
    lv = torch.tensor([0, 1, 2], dtype=torch.int16)
    assert D._walk_order(cyc, lv) is lv


def test_pair_bitmap_and_tile_list_of_the_dense_self_collision_entry():
    """backends/geometry.pair_bitmap: bit (j % 32) of word [j / 32, i] per listed pair, the 16 x 16 tiles that hold a
    pair as (i / 16) | (j / 16) << 8, None for lists the dense kernel's tie rule does not cover."""
    import numpy as np
    import torch

    from curobo_amd.backends import geometry as G

    S = 70
    pairs = [(0, 1), (0, 33), (2, 69), (17, 40), (40, 41)]
    t = torch.tensor(pairs, dtype=torch.int16)
    bm, nslots, tiles, masks = G.pair_bitmap(t, S)
    assert nslots == 4 and tuple(bm.shape) == (8, 256)
    # the lane masks: bit reg of masks[c, lane] <-> pair (16 ib + 4 (lane / 16) + reg, 16 jb + lane % 16) of tile c
    assert tuple(masks.shape) == (tiles.shape[0], 64) and masks.dtype == torch.uint8
    listed = set()
    for c, tl in enumerate(tiles.tolist()):
        ib, jb = tl & 0xff, tl >> 8
        for lane in range(64):
            for reg in range(4):
                if (int(masks[c, lane]) >> reg) & 1:
                    listed.add((16 * ib + 4 * (lane // 16) + reg, 16 * jb + lane % 16))
    assert listed == set(pairs)
    words = bm.numpy().view(np.uint32)
    want = np.zeros_like(words)
    for i, j in pairs:
        want[j // 32, i] |= np.uint32(1) << np.uint32(j % 32)
    assert np.array_equal(words, want)
    assert sorted(tiles.tolist()) == sorted({(i // 16) | ((j // 16) << 8) for i, j in pairs})
    assert G.pair_bitmap(t, S) is G.pair_bitmap(t, S)
    unsorted = torch.tensor([(0, 33), (0, 1)], dtype=torch.int16)
    assert G.pair_bitmap(unsorted, S) is None
    swapped = torch.tensor([(5, 2)], dtype=torch.int16)
    assert G.pair_bitmap(swapped, S) is None


def test_self_lane_lists_host_deals_every_pair_once():
    """curobo_hip_self_lane_lists_host (pure host code): every pair of the packaged robots lands in the list of exactly one of
    its two spheres, lists are as short as a perfect balance allows (+1), padding points at the NaN sphere; robots
    outside the form return 0"""
    import numpy as np
    import torch

    from curobo_amd.backends.rollout import attach_self_lane_lists
    from curobo_amd.robot import load_packaged_robot

    for name, expect in (("franka", True), ("ur10e", True), ("unitree_g1", False)):
        m = load_packaged_robot(name)
        pairs = torch.as_tensor(np.asarray(m.collision_pairs)).to(torch.int16).contiguous()
        attach_self_lane_lists(pairs, m.num_spheres)
        got = getattr(pairs, "_self_lane_lists", None)
        assert (got is not None) == expect, name
        if got is None:
            continue
        words, code = got
        len0, len1 = code & 0xffff, code >> 16
        S, P = m.num_spheres, pairs.shape[0]
        owners = min(S, 64)
        assert len1 == 0 and len0 <= -(-P // owners) + 1
        w = words.numpy().astype(np.uint32).reshape(len0 + len1, 64)
        pad = np.uint32((S * 16) | (0xffff << 16))
        live = w != pad
        k = (w >> 16)[live]
        assert np.array_equal(np.sort(k), np.arange(P))
        lane = np.broadcast_to(np.arange(64), w.shape)[live]
        partner = ((w & 0xffff) // 16)[live]
        pn = pairs.numpy()
        assert all({int(a), int(b)} == {int(pn[kk, 0]), int(pn[kk, 1])} for a, b, kk in zip(lane, partner, k))
