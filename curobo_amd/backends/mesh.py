"""HIP backend for triangle-mesh obstacles: linear BVH build, closest-point / sign query, sphere-vs-mesh collision and the
ESDF bake through the BVH (``csrc/mesh_bvh.hip``).

The reference queries meshes through NVIDIA Warp (``curobo/_src/geom/data/data_mesh.py:555-700``: ``wp.mesh_query_point``
per query sphere; ``MeshData`` :60-120 holds the Warp mesh handles, poses, enable flags); Warp has no ROCm backend and is
not in the reference tree, so what these functions follow is the contract of that query as the reference uses it.
"""

from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
import os
from typing import Optional, Tuple

import numpy as np
import torch

from .._lib import check, current_stream, load, ptr


class Mesh(C.Structure):
    """``curobo_hip_mesh``"""

    _fields_ = [("tri", C.c_void_p), ("node_box", C.c_void_p), ("tri_pn", C.c_void_p), ("n_tri", C.c_int32), ("n_leaves", C.c_int32),
                ("leaf_size", C.c_int32), ("sign_rule", C.c_int32),
                ("cell_start", C.c_void_p), ("cell_list", C.c_void_p), ("grid_lo", C.c_float * 3), ("grid_h", C.c_float),
                ("grid_n", C.c_int32 * 3), ("grid_pad", C.c_float)]


class MeshSet(C.Structure):
    """``curobo_hip_mesh_set``"""

    _fields_ = [("meshes", C.c_void_p), ("mesh_id", C.c_void_p), ("dims", C.c_void_p), ("inv_pose", C.c_void_p),
                ("enable", C.c_void_p), ("count", C.c_void_p), ("max_n", C.c_int32), ("gradient_mode", C.c_int32),
                ("num_envs", C.c_int32), ("flags", C.c_int32)]


MESH_SET_HAS_CELLS = 1  # curobo_hip_mesh_set.flags


@dataclass
class DeviceMesh:
    """one mesh on the device: sorted triangles + node boxes (kept alive here), the C struct, its bounding box"""

    tri: torch.Tensor
    node_box: torch.Tensor
    tri_pn: torch.Tensor
    struct: Mesh
    bounds: np.ndarray  # [2, 3] lo / hi in the mesh frame
    n_tri: int
    #: SIGN_CLOSEST_FEATURE (closed, consistently oriented) or SIGN_WARP_RAYS (anything else: the reference's three rays)
    sign_rule: int = 0
    #: the cell lists (``build_mesh_cells``): packed cell words, (triangle, distance) entries -- kept alive here -- and what was built
    cell_start: Optional[torch.Tensor] = None
    cell_list: Optional[torch.Tensor] = None
    cells_info: Optional[dict] = None

    @property
    def dims(self) -> np.ndarray:
        return (self.bounds[1] - self.bounds[0]).astype(np.float32)


def feature_pseudonormals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """[F, 6, 4] float32: per triangle the angle-weighted pseudonormals of its vertices a, b, c and the pseudonormals of its
    edges ab, bc, ca (sum of the unit normals of the faces sharing the edge) -- Baerentzen & Aanaes, "Signed distance
    computation using the angle weighted pseudonormal" (2005): for a closed, consistently oriented surface the sign of
    (p - closest point) . pseudonormal of the feature the closest point lies on is the sign of the signed distance.  Vertices
    are welded by position first (faces of a mesh file often do not share indices along creases).  Scene-upload plumbing,
    once per loaded mesh; a feature whose pseudonormal degenerates (|n| ~ 0) is left zero: the kernels then count crossings."""
    v = np.asarray(vertices, np.float64)
    f = np.asarray(faces, np.int64)
    span = float(np.ptp(v, axis=0).max()) or 1.0
    key = np.round((v - v.min(0)) / (span * 1e-7)).astype(np.int64)
    _, weld = np.unique(key, axis=0, return_inverse=True)
    weld = weld.reshape(-1)
    fw = weld[f]                                     # faces over welded vertex ids
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-300), 0.0)

    def angle(p, q, r):  # angle at p
        u, w = q - p, r - p
        cu = np.einsum("ij,ij->i", u, w) / np.maximum(np.linalg.norm(u, axis=1) * np.linalg.norm(w, axis=1), 1e-300)
        return np.arccos(np.clip(cu, -1.0, 1.0))
    ang = np.stack([angle(a, b, c), angle(b, c, a), angle(c, a, b)], axis=1)           # [F, 3]
    nv = np.zeros((int(weld.max()) + 1, 3))
    for k in range(3):
        np.add.at(nv, fw[:, k], ang[:, k:k + 1] * n)
    out = np.zeros((f.shape[0], 6, 4), np.float32)
    out[:, 0:3, :3] = nv[fw]                                                            # vertices a, b, c
    # edges ab, bc, ca: undirected key over welded ids
    e0 = np.stack([fw[:, [0, 1]], fw[:, [1, 2]], fw[:, [2, 0]]], axis=1)                # [F, 3, 2]
    ek = np.sort(e0.reshape(-1, 2), axis=1)
    _, inv = np.unique(ek, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    ne = np.zeros((int(inv.max()) + 1, 3))
    np.add.at(ne, inv, np.repeat(n, 3, axis=0))
    out[:, 3:6, :3] = ne[inv].reshape(-1, 3, 3)
    return out


#: how the sign of a mesh query is decided (``curobo_hip_mesh.sign_rule``)
SIGN_CLOSEST_FEATURE, SIGN_WARP_RAYS = 0, 1


def _weld_ids(vertices: np.ndarray) -> np.ndarray:
    v = np.asarray(vertices, np.float64)
    span = float(np.ptp(v, axis=0).max()) or 1.0
    key = np.round((v - v.min(0)) / (span * 1e-7)).astype(np.int64)
    _, weld = np.unique(key, axis=0, return_inverse=True)
    return weld.reshape(-1)


def mesh_is_closed_and_oriented(vertices: np.ndarray, faces: np.ndarray) -> bool:
    """True when, over vertices welded by position, every directed edge occurs exactly once and its reverse exactly once: a closed
    2-manifold whose faces all wind the same way.  On such a surface the reference's sign -- Warp's ``mesh_query_point``: rays along
    +x, +y, +z, inside iff every ray's nearest hit is a back face (warp/native/mesh.h ``mesh_query_inside``) -- is the same function
    of the query point as the closest-feature pseudonormal sign the kernels evaluate without a ray; on anything else (an open
    scan, a flipped face, a T-junction) the two differ and the kernels cast Warp's three rays (``SIGN_WARP_RAYS``)."""
    f = _weld_ids(vertices)[np.asarray(faces, np.int64)]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    if (e[:, 0] == e[:, 1]).any():
        return False
    key = e[:, 0] * (int(f.max()) + 1) + e[:, 1]
    rkey = e[:, 1] * (int(f.max()) + 1) + e[:, 0]
    uk, cnt = np.unique(key, return_counts=True)
    return bool((cnt == 1).all() and np.isin(rkey, uk).all())


def build_mesh_bvh(vertices, faces, device, leaf_size: int = 8, sign_rule: Optional[int] = None, cells=None) -> DeviceMesh:
    """vertices [V, 3] (mesh frame), faces [F, 3] -> the linear BVH on ``device``: Morton keys of the centroids (HIP), the
    sort (torch: plumbing), triangles in sorted order + node boxes (HIP, one launch per level).  ``sign_rule``: None = by the
    mesh's topology (``mesh_is_closed_and_oriented``: closest feature on closed consistently oriented meshes, the reference's
    three rays on anything else), or forced.  ``cells``: build the cell lists too (None: on unless CUROBO_MESH_CELLS=0; a dict =
    the keyword arguments of ``build_mesh_cells``)."""
    try:  # development knob (a malformed value is ignored)
        leaf_size = max(1, int(os.environ.get("CUROBO_MESH_LEAF_SIZE", leaf_size)))
    except ValueError:
        pass
    v = torch.as_tensor(np.ascontiguousarray(vertices, np.float32)).to(device).contiguous()
    f = torch.as_tensor(np.ascontiguousarray(faces, np.int32)).to(device).contiguous()
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3 or f.shape[0] == 0:
        raise ValueError(f"mesh needs vertices [V, 3] and faces [F, 3], got {tuple(v.shape)} and {tuple(f.shape)}")
    if int(f.min()) < 0 or int(f.max()) >= v.shape[0]:
        raise ValueError("face indices out of range")
    n = int(f.shape[0])
    used = v[f.reshape(-1).long()]
    bounds = np.stack([used.min(0).values.cpu().numpy(), used.max(0).values.cpu().numpy()]).astype(np.float32)
    codes = torch.empty(n, dtype=torch.int64, device=device)
    b = (C.c_float * 6)(*[float(x) for x in bounds.reshape(-1)])
    check(load().curobo_hip_mesh_morton_codes(ptr(codes), ptr(v), ptr(f), n, C.cast(b, C.c_void_p), current_stream(v)))
    codes = torch.sort(codes).values.contiguous()
    n_leaves = 1
    while n_leaves * leaf_size < n:
        n_leaves *= 2
    tri = torch.zeros(n, 12, device=device)
    box = torch.zeros(2 * n_leaves, 8, device=device)
    check(load().curobo_hip_mesh_bvh_build(ptr(tri), ptr(box), ptr(v), ptr(f), ptr(codes), n, n_leaves, leaf_size, current_stream(v)))
    # pseudonormals of the triangle features, in the sorted order of the triangles (the low half of a key is the index)
    pn = torch.as_tensor(feature_pseudonormals(np.asarray(vertices, np.float32), np.asarray(faces, np.int64))).to(device)
    tri_pn = pn[(codes & 0xFFFFFFFF).long()].contiguous()
    rule = SIGN_CLOSEST_FEATURE if sign_rule is None and mesh_is_closed_and_oriented(vertices, faces) else \
        (SIGN_WARP_RAYS if sign_rule is None else int(sign_rule))
    s = Mesh(ptr(tri), ptr(box), ptr(tri_pn), n, n_leaves, leaf_size, rule)
    mesh = DeviceMesh(tri, box, tri_pn, s, bounds, n, rule)
    if cells is None:
        cells = os.environ.get("CUROBO_MESH_CELLS", "1") != "0"
    if cells:
        build_mesh_cells(mesh, **(cells if isinstance(cells, dict) else {}))
    return mesh


def build_mesh_cells(mesh: DeviceMesh, cell_size: Optional[float] = None, pad: float = 0.2, max_cells: int = 1 << 18,
                     gather_cap: int = 2048, max_entries: int = 1 << 26) -> DeviceMesh:
    """The distance-sorted closest-triangle cell lists of ``mesh`` (``curobo_hip_mesh.cell_start / cell_list``; the query is
    ``csrc/mesh_device.hpp::mesh_cells_sdf``), written into ``mesh.struct`` in place.

    A uniform grid over the bounding box grown by ``pad`` (how far from the box a query can still be answered without a tree
    walk: the largest sphere radius + activation distance + half a sweep step one expects; 0.2 m: measured on the bench's mesh
    world, 0.1 m sends 2 % of the live spheres -- the fast-moving ones -- to the tree walk and 0.2 m none); cells of ``cell_size``
    (default 2 cm), coarsened until the grid has at most ``max_cells`` cells.  Two launches around a prefix sum and a sort (torch:
    plumbing): count -> offsets -> fill + keys -> sort -> gather.  ``gather_cap``: cells whose candidate ball holds more
    triangles get no list (their queries walk the tree).  ``max_entries`` (2^26 entries = 1 GiB): the budget of one mesh's lists."""
    lib = load()
    dev = mesh.tri.device
    lo, hi = mesh.bounds[0].astype(np.float64) - pad, mesh.bounds[1].astype(np.float64) + pad
    h = float(cell_size) if cell_size else float(os.environ.get("CUROBO_MESH_CELL_SIZE", 0.02))
    ext = hi - lo
    st = mesh.struct
    st.cell_start, st.cell_list = None, None
    mesh.cell_start = mesh.cell_list = mesh.cells_info = None
    # The lists' size is known only after the count pass.  A dense mesh (entries per cell grow with the triangle density, the
    # total falls with the square of the cell size) gets coarser cells until the lists fit ``max_entries``; a mesh that does not
    # fit after a few tries keeps the tree walk alone (a slower launch, the same results).
    for attempt in range(5):
        while np.prod(np.ceil(ext / h)) > max_cells:
            h *= 1.1
        n3 = np.maximum(np.ceil(ext / h).astype(np.int64), 1)
        for i in range(3):
            st.grid_lo[i] = float(lo[i])
            st.grid_n[i] = int(n3[i])
        st.grid_h, st.grid_pad = h, float(pad)
        n_cells = int(np.prod(n3))
        count = torch.empty(n_cells, dtype=torch.int32, device=dev)
        cover = torch.empty(n_cells, dtype=torch.float32, device=dev)
        side = torch.empty(n_cells, dtype=torch.uint8, device=dev)
        centre_dist = torch.empty(n_cells, dtype=torch.float32, device=dev)
        stream = current_stream(count)
        check(lib.curobo_hip_mesh_cells_count(ptr(count), ptr(cover), ptr(side), ptr(centre_dist), C.addressof(st), int(gather_cap), stream))
        offsets = torch.zeros(n_cells + 1, dtype=torch.int64, device=dev)
        torch.cumsum(count, 0, out=offsets[1:])
        total = int(offsets[-1])
        if total <= max_entries:
            break
        h *= 1.5
    else:
        st.grid_h = 0.0
        mesh.cells_info = {"skipped": f"{total} list entries at a cell size of {h / 1.5:.3f} m exceed max_entries = {max_entries}: tree walk only"}
        return mesh
    keys = torch.empty(total, dtype=torch.int64, device=dev)
    entries = torch.empty(total, 4, dtype=torch.int32, device=dev)
    cell_start = torch.empty(n_cells + 1, 2, dtype=torch.int32, device=dev)
    check(lib.curobo_hip_mesh_cells_fill(ptr(keys), ptr(entries), ptr(cell_start), ptr(offsets), ptr(cover), ptr(side), ptr(centre_dist),
                                         C.addressof(st), stream))
    perm = torch.sort(keys).indices
    mesh.cell_list = entries[perm].contiguous()
    mesh.cell_start = cell_start
    st.cell_start, st.cell_list = ptr(mesh.cell_start), ptr(mesh.cell_list)
    listed = count > 1
    mesh.cells_info = {"cell_size": h, "grid": [int(v) for v in n3], "pad": float(pad), "cells": n_cells, "entries": total,
                       "bytes": int(total * 16 + (n_cells + 1) * 8), "cells_without_list": int((cover == 0).sum()),
                       "mean_list": float(count[listed].float().mean() - 1) if bool(listed.any()) else 0.0,
                       "max_list": int(count.max()) - 1}
    return mesh


def mesh_query(mesh: DeviceMesh, points: torch.Tensor, max_distance: float, want_grad: bool = True
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """points [n, 3] in the mesh frame -> (sdf [n], gradient [n, 3]) as ``compute_local_sdf_with_grad`` returns them"""
    p = points.to(torch.float32).contiguous()
    n = int(p.shape[0])
    sdf = torch.empty(n, device=p.device)
    grad = torch.empty(n, 3, device=p.device) if want_grad else None
    check(load().curobo_hip_mesh_query(ptr(sdf), ptr(grad), ptr(p), C.addressof(mesh.struct), float(max_distance), n, current_stream(p)))
    return sdf, grad


def mesh_esdf_bake_bvh(out_esdf: torch.Tensor, mesh: DeviceMesh, grid_shape, voxel_size: float, grid_to_mesh_3x4,
                       max_distance: float = 100.0) -> torch.Tensor:
    assert out_esdf.dtype == torch.float16 and out_esdf.is_contiguous()
    nx, ny, nz = (int(v) for v in grid_shape)
    assert out_esdf.numel() == nx * ny * nz
    xf = (C.c_float * 12)(*[float(v) for v in grid_to_mesh_3x4])
    check(load().curobo_hip_mesh_esdf_bake_bvh(ptr(out_esdf), C.addressof(mesh.struct), nx, ny, nz, float(voxel_size), float(max_distance),
                                               C.cast(xf, C.c_void_p), current_stream(out_esdf)))
    return out_esdf


def sphere_mesh_collision(distance, gradient, spheres, mesh_set: MeshSet, weight, activation_distance, env_query_idx, batch_size: int,
                          horizon: int, num_spheres: int, use_multi_env: bool, sweep_steps: int = 0, enable_speed_metric: bool = False,
                          speed_dt=None, accumulate: bool = True, workspace=None):
    """the mesh share of the scene-collision forward (``curobo_hip_sphere_mesh_collision_ws``: survivors of the bounding-box
    reject queued launch-wide, see include/curobo_hip.h; ``workspace=False`` runs the one-kernel form).  Without a caller-owned
    ``workspace`` one is kept per OUTPUT BUFFER (device, size, address of ``distance``): the counter and queue of a launch
    belong to the rollout that owns the output, so rollouts of equal size that run concurrently -- the seed shards of
    ``PipelinedLBFGS`` on their own streams, the parallel branches of one captured graph -- never share them (a cache keyed
    by size alone let one shard's select kernel clear or fill the queue another shard's walk kernel was reading); launches
    into one output buffer are ordered by its owner's stream; the workspace is held BY the output tensor (an attribute of it, or
    of its base for a view), so a captured graph's workspace lives as long as the buffer the graph writes."""
    lib = load()
    if workspace is False:
        check(lib.curobo_hip_sphere_mesh_collision(
            ptr(distance), ptr(gradient), ptr(spheres), C.addressof(mesh_set), ptr(weight), ptr(activation_distance), ptr(env_query_idx),
            batch_size, horizon, num_spheres, int(use_multi_env), sweep_steps, int(enable_speed_metric), ptr(speed_dt), int(accumulate),
            current_stream(distance)))
        return
    nbytes = C.c_int64(0)
    check(lib.curobo_hip_sphere_mesh_collision_ws_bytes(batch_size, horizon, num_spheres, C.cast(C.pointer(nbytes), C.c_void_p)))
    need = int(nbytes.value)
    if workspace is None:
        # owned by the OUTPUT tensor (its base when a view is handed in): the workspace lives exactly as long as the buffer
        # whose launches it serves -- nothing piles up when rollouts are rebuilt, and a freed address that comes back for
        # another rollout's buffer never inherits a workspace an older captured graph may still point at
        owner = distance._base if distance._base is not None else distance
        held = getattr(owner, "_curobo_mesh_ws", None)
        if held is None:
            held = owner._curobo_mesh_ws = {}
        key = (distance.device, need, int(distance.data_ptr()))
        workspace = held.get(key)
        if workspace is None:
            workspace = held[key] = torch.empty(need, dtype=torch.uint8, device=distance.device)
            _WORKSPACES[key] = workspace
    check(lib.curobo_hip_sphere_mesh_collision_ws(
        ptr(distance), ptr(gradient), ptr(spheres), C.addressof(mesh_set), ptr(weight), ptr(activation_distance), ptr(env_query_idx),
        batch_size, horizon, num_spheres, int(use_multi_env), sweep_steps, int(enable_speed_metric), ptr(speed_dt), int(accumulate),
        ptr(workspace), int(workspace.numel()), current_stream(distance)))


#: the live launch workspaces, for inspection only (weak: an entry goes when the output tensor that owns it does)
_WORKSPACES = weakref.WeakValueDictionary()
