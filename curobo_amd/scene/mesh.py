"""Mesh obstacles of a scene: the device-side store the mesh collision kernel reads.

Counterpart of the reference's ``MeshData`` (``curobo/_src/geom/data/data_mesh.py:60-440``: a cache of loaded meshes, and
per environment slot a mesh handle, inverse pose, bounding-box dims, enable flag; ``Mesh`` obstacles of a ``SceneCfg``,
``geom/types.py``).  A mesh is loaded once (``build_mesh_bvh``: linear BVH on the device) and may be placed any number of
times; poses and enable flags are updated in place.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import os

import numpy as np
import torch

from ..backends.mesh import MESH_SET_HAS_CELLS, DeviceMesh, Mesh, MeshSet, build_mesh_bvh
from .data import inverse_pose7


def load_obj(path: str):
    """(vertices [V, 3], faces [F, 3]) of a Wavefront OBJ file (v / f records; polygons are fan-triangulated)"""
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) for x in t[1:]]
                idx = [i - 1 if i > 0 else len(v) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    f.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def _weld(tri: np.ndarray):
    """[F, 3, 3] triangle corners -> (vertices [V, 3], faces [F, 3]) with identical corners shared (STL stores every corner per triangle)"""
    v, inverse = np.unique(tri.reshape(-1, 3), axis=0, return_inverse=True)
    return v.astype(np.float32), inverse.reshape(-1, 3).astype(np.int32)


def load_stl(path: str):
    """(vertices [V, 3], faces [F, 3]) of an STL file, binary (80-byte header, uint32 count, 50-byte records) or ASCII (``vertex x y z``
    lines); corners that coincide exactly are welded, degenerate triangles dropped"""
    with open(path, "rb") as fh:
        raw = fh.read()
    n = int(np.frombuffer(raw[80:84], "<u4")[0]) if len(raw) >= 84 else -1
    if n >= 0 and len(raw) == 84 + 50 * n:  # the binary layout is decided by the size, not by a header that may start with "solid"
        rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
        tri = rec["v"].astype(np.float32)
    else:
        pts = [[float(x) for x in line.split()[1:4]] for line in raw.decode("ascii", "ignore").splitlines() if line.strip().startswith("vertex")]
        if not pts or len(pts) % 3:
            raise ValueError(f"{path}: not an STL file this loader reads ({len(pts)} vertex records)")
        tri = np.asarray(pts, np.float32).reshape(-1, 3, 3)
    v, f = _weld(tri)
    keep = (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])
    return v, f[keep]


def load_mesh_file(path: str):
    """(vertices, faces) of a mesh file by its extension: Wavefront OBJ or STL (the formats of the reference's robot and scene assets that
    need no third-party reader; it reads everything through ``trimesh``)"""
    ext = path.lower().rsplit(".", 1)[-1]
    if ext == "obj":
        return load_obj(path)
    if ext == "stl":
        return load_stl(path)
    raise ValueError(f"{path}: mesh files are read as .obj or .stl (give ``vertices`` and ``faces`` for anything else)")


class MeshStore:
    """``envs[e]`` = list of ``{"name": str, "vertices": [V, 3], "faces": [F, 3] (or "file_path": *.obj / *.stl), "pose": [x, y, z,
    qw, qx, qy, qz], "scale": [sx, sy, sz] (optional), "enable": bool}``; meshes that share a name share one BVH."""

    #: 0 = the local gradient as the reference's mesh query returns it (data_mesh.py:693-697: away from the surface for a centre
    #: outside it -- the opposite of what its cuboid and voxel queries return there); 1 = toward the obstacle on both sides
    REFERENCE_GRADIENT, CONSISTENT_GRADIENT = 0, 1

    def __init__(self, envs: List[List[Dict]], device, max_n: Optional[int] = None, leaf_size: int = 8,
                 gradient_mode: Optional[int] = None, cells=None, sign_rule: Optional[int] = None):
        self.device = torch.device(device)
        if gradient_mode is None:  # (``CUROBO_MESH_GRADIENT`` = reference | consistent: the default of stores that are not told)
            gradient_mode = {"reference": self.REFERENCE_GRADIENT, "consistent": self.CONSISTENT_GRADIENT}[
                os.environ.get("CUROBO_MESH_GRADIENT", "reference")]
        self.gradient_mode = int(gradient_mode)
        E = len(envs)
        n = max_n or max(1, max(len(e) for e in envs))
        self.cache: Dict[str, int] = {}
        self.meshes: List[DeviceMesh] = []
        mesh_id = np.zeros((E, n), np.int32)
        dims = np.zeros((E, n, 4), np.float32)
        inv_pose = np.zeros((E, n, 8), np.float32)
        inv_pose[..., 3] = 1.0
        enable = np.zeros((E, n), np.uint8)
        count = np.zeros((E,), np.int32)
        self.names: List[List[Optional[str]]] = [[None] * n for _ in range(E)]
        for e, obs in enumerate(envs):
            count[e] = len(obs)
            for i, o in enumerate(obs):
                name = o.get("name", f"mesh_{e}_{i}")
                key = o.get("mesh_name", name)
                if key not in self.cache:
                    if "vertices" in o:
                        v, f = np.asarray(o["vertices"], np.float32), np.asarray(o["faces"], np.int32)
                    else:
                        v, f = load_mesh_file(o["file_path"])
                    if o.get("scale") is not None:
                        v = v * np.asarray(o["scale"], np.float32).reshape(1, 3)
                    self.cache[key] = len(self.meshes)
                    self.meshes.append(build_mesh_bvh(v, f, self.device, leaf_size, sign_rule=o.get("sign_rule", sign_rule), cells=cells))
                mid = self.cache[key]
                mesh_id[e, i] = mid
                dims[e, i, :3] = self.meshes[mid].dims
                inv_pose[e, i, :7] = inverse_pose7(o["pose"])
                enable[e, i] = 1 if o.get("enable", True) else 0
                self.names[e][i] = name
        t = lambda a: torch.as_tensor(a).to(self.device).contiguous()  # noqa: E731
        self.mesh_id, self.dims, self.inv_pose, self.enable, self.count = t(mesh_id), t(dims), t(inv_pose), t(enable), t(count)
        raw = b"".join(bytes(m.struct) for m in self.meshes)
        self._mesh_structs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device).contiguous()
        self.max_n, self.num_envs = n, E
        self.struct = MeshSet(self._mesh_structs.data_ptr(), self.mesh_id.data_ptr(), self.dims.data_ptr(), self.inv_pose.data_ptr(),
                              self.enable.data_ptr(), self.count.data_ptr(), n, int(gradient_mode), E,
                              MESH_SET_HAS_CELLS if any(m.cell_start is not None for m in self.meshes) else 0)
        self.envs = envs

    def _slot(self, name: str, env_idx: int) -> int:
        try:
            return self.names[env_idx].index(name)
        except ValueError:
            raise ValueError(f"Mesh with name '{name}' not found in environment {env_idx}") from None

    def update_pose(self, name: str, pose7: Sequence[float], env_idx: int = 0) -> None:
        """reference MeshData.update_pose: the obstacle moves, the BVH (mesh frame) stays"""
        i = self._slot(name, env_idx)
        self.inv_pose[env_idx, i, :7] = torch.as_tensor(inverse_pose7(pose7), dtype=torch.float32, device=self.device)

    def set_enabled(self, name: str, enabled: bool, env_idx: int = 0) -> None:
        self.enable[env_idx, self._slot(name, env_idx)] = int(enabled)
