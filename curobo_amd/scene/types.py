"""Obstacle and scene descriptions: the reference's ``curobo.scene`` types (``curobo/_src/geom/types.py``: ``Obstacle`` :37-62,
``Cuboid`` :254-264, ``Capsule`` :289-300, ``Cylinder`` :339-347, ``Sphere`` :372-388, ``Mesh`` :451-488, ``VoxelGrid`` :809-846,
``SceneCfg`` :918-1292 -- exported as ``Scene``).  Only what describes collision geometry is mirrored: the fields, ``SceneCfg.create``
from the yaml / dictionary format, ``add_obstacle`` / ``remove_obstacle`` / ``get_obstacle``, and the oriented-bounding-box
approximation of the analytic kinds.  Rendering members (materials, textures, trimesh exports) are not.

``scene_from_config`` (scene/config.py) turns a ``SceneCfg`` -- or a list of them, one per environment -- into the obstacle stores
the collision kernels read."""

from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np


@dataclass
class Obstacle:
    #: unique name
    name: str
    #: [x, y, z, qw, qx, qy, qz]
    pose: Optional[List[float]] = None
    #: per-axis scale (meshes only)
    scale: Optional[List[float]] = None
    #: rgba (kept for round trips; nothing here renders)
    color: Optional[List[float]] = None
    #: obstacles can be switched off without leaving the scene (this package's stores carry an enable flag per slot)
    enable: bool = True

    def _need_pose(self) -> List[float]:
        if self.pose is None:
            raise ValueError(f"{type(self).__name__} obstacle '{self.name}' requires a pose")
        return [float(x) for x in self.pose]

    def get_transform_matrix(self) -> np.ndarray:
        """the pose as a homogeneous 4 x 4 matrix (reference ``Obstacle.get_transform_matrix``, geom/types.py:160-170)"""
        P = Pose7(self._need_pose())
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = P.R, P.t
        return m

    def get_sphere(self, n: int = 1) -> "Sphere":
        """one sphere at the centre of the obstacle's bounding cuboid with radius = the cuboid's smallest edge, as the reference
        builds it (``Obstacle.get_sphere``, geom/types.py:172-194: the radius is the edge, not half of it)"""
        obb = self.get_cuboid()
        return Sphere(name="m_sphere", pose=list(obb.pose), radius=float(min(obb.dims)))


@dataclass
class Cuboid(Obstacle):
    #: edge lengths [x, y, z] in metres
    dims: Sequence[float] = (0.0, 0.0, 0.0)

    def __post_init__(self):
        self._need_pose()

    def get_cuboid(self) -> "Cuboid":
        return self


@dataclass
class Capsule(Obstacle):
    radius: float = 0.0
    #: end points of the axis in the obstacle frame
    base: Sequence[float] = (0.0, 0.0, 0.0)
    tip: Sequence[float] = (0.0, 0.0, 0.0)

    def get_cuboid(self) -> Cuboid:
        """the box around the capsule, in the capsule's frame (axis-aligned there: exact for an axis along z)"""
        b, t, r = np.asarray(self.base, np.float64), np.asarray(self.tip, np.float64), float(self.radius)
        lo, hi = np.minimum(b, t) - r, np.maximum(b, t) + r
        centre = Pose7(self._need_pose()).transform(0.5 * (lo + hi))
        return Cuboid(name=self.name, pose=list(centre) + list(self._need_pose()[3:]), dims=list(hi - lo), color=self.color, enable=self.enable)


@dataclass
class Cylinder(Obstacle):
    radius: float = 0.0
    #: along z of the obstacle frame, centred
    height: float = 0.0

    def get_cuboid(self) -> Cuboid:
        d = 2.0 * float(self.radius)
        return Cuboid(name=self.name, pose=self._need_pose(), dims=[d, d, float(self.height)], color=self.color, enable=self.enable)


@dataclass
class Sphere(Obstacle):
    radius: float = 0.0
    #: deprecated in the reference: use ``pose``
    position: Optional[List[float]] = None

    def __post_init__(self):
        if self.position is not None:
            self.pose = list(self.position) + [1, 0, 0, 0]
        if self.pose is not None:
            self.position = list(self.pose[:3])

    def get_cuboid(self) -> Cuboid:
        d = 2.0 * float(self.radius)
        return Cuboid(name=self.name, pose=self._need_pose(), dims=[d, d, d], color=self.color, enable=self.enable)


@dataclass
class Mesh(Obstacle):
    #: Wavefront OBJ or STL file (``scene.mesh.load_mesh_file``), or ``vertices`` [V, 3] + ``faces`` [F, 3]
    file_path: Optional[str] = None
    vertices: Optional[Any] = None
    faces: Optional[Any] = None

    def __post_init__(self):
        if self.scale is not None and self.vertices is not None:  # as the reference: scaled once, here
            self.vertices = np.asarray(self.vertices, np.float32) * np.ravel(np.asarray(self.scale, np.float32))
            self.scale = None

    def get_mesh_data(self) -> Tuple[np.ndarray, np.ndarray]:
        if self.vertices is not None:
            return np.asarray(self.vertices, np.float32), np.asarray(self.faces, np.int32).reshape(-1, 3)
        from .mesh import load_mesh_file

        v, f = load_mesh_file(self.file_path)
        if self.scale is not None:
            v = v * np.ravel(np.asarray(self.scale, np.float32))
        return v, f

    def get_cuboid(self) -> Cuboid:
        """the box around the vertices, axis-aligned in the mesh frame"""
        v, _ = self.get_mesh_data()
        lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
        centre = Pose7(self._need_pose()).transform(0.5 * (lo + hi))
        return Cuboid(name=self.name, pose=list(centre) + list(self._need_pose()[3:]), dims=list(hi - lo), color=self.color, enable=self.enable)


@dataclass
class VoxelGrid(Obstacle):
    """an ESDF sampled on a regular grid (positive inside obstacles, as the reference's)"""

    #: extent [x, y, z] in metres; the grid has round(dims / voxel_size) cells per axis
    dims: Sequence[float] = (0.0, 0.0, 0.0)
    voxel_size: float = 0.02
    #: ESDF values, [nx * ny * nz] (x slowest) or [nx, ny, nz]
    feature_tensor: Optional[Any] = None

    #: [n, 4] cell centres (x y z 0) as ``create_xyzr_tensor`` lays them out (x slowest): the order of ``feature_tensor``
    xyzr_tensor: Optional[Any] = None

    def get_grid_shape(self) -> Tuple[List[int], List[float], List[float]]:
        shape = [int(round(float(x) / self.voxel_size)) for x in self.dims]
        half = [0.5 * float(x) for x in self.dims]
        return shape, [-h for h in half], half

    def create_xyzr_tensor(self, transform_to_origin: bool = False, device_cfg=None):
        """cell centres of the grid [nx * ny * nz, 4] (x slowest, r = 0) in the grid's frame, or in the world's with
        ``transform_to_origin`` (reference ``VoxelGrid.create_xyzr_tensor``, geom/types.py:846-883: centre i of an axis of n cells
        = (i + 1 - round(0.5 extent / voxel_size)) voxel_size - 0.5 voxel_size)"""
        import torch

        dev = device_cfg.device if device_cfg is not None else "cpu"
        n, _, _ = self.get_grid_shape()
        inv = 1.0 / self.voxel_size
        axes = [(torch.linspace(1, n[a], n[a], device=dev) - round(0.5 * float(self.dims[a]) * inv)) * self.voxel_size - 0.5 * self.voxel_size
                for a in range(3)]
        xyz = torch.stack(torch.meshgrid(*axes, indexing="ij")).permute(1, 2, 3, 0).reshape(-1, 3)
        if transform_to_origin:
            xyz = torch.as_tensor(Pose7(self._need_pose()).transform(xyz.cpu().numpy()), dtype=xyz.dtype, device=xyz.device)
        return torch.cat([xyz, torch.zeros_like(xyz[:, :1])], dim=1)

    def get_occupied_voxels(self, feature_threshold: Optional[float] = None):
        """[m, 4] centres + ESDF value of the cells whose value exceeds the threshold (default: half a voxel inside; reference :885-901)"""
        if feature_threshold is None:
            feature_threshold = -0.5 * self.voxel_size
        if self.xyzr_tensor is None or self.feature_tensor is None:
            raise ValueError("Feature tensor or xyzr tensor is empty")
        xyzr = self.xyzr_tensor.clone()
        xyzr[:, 3] = self.feature_tensor.reshape(-1).to(xyzr.dtype)
        return xyzr[self.feature_tensor.reshape(-1) > feature_threshold]

    def clone(self) -> "VoxelGrid":
        c = lambda t: t.clone() if hasattr(t, "clone") else copy.deepcopy(t)  # noqa: E731
        return VoxelGrid(name=self.name, pose=list(self.pose) if self.pose is not None else None, dims=list(self.dims), voxel_size=self.voxel_size,
                         feature_tensor=None if self.feature_tensor is None else c(self.feature_tensor),
                         xyzr_tensor=None if self.xyzr_tensor is None else c(self.xyzr_tensor), enable=self.enable)


class Pose7:
    """[x, y, z, qw, qx, qy, qz] acting on points (host side, float64)"""

    def __init__(self, pose: Sequence[float]):
        p = np.asarray(pose, np.float64)
        self.t, q = p[:3], p[3:7] / np.linalg.norm(p[3:7])
        w, x, y, z = q
        self.R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def transform(self, p) -> np.ndarray:
        return np.asarray(p, np.float64) @ self.R.T + self.t


_KINDS = (("sphere", Sphere), ("cuboid", Cuboid), ("capsule", Capsule), ("mesh", Mesh), ("cylinder", Cylinder), ("voxel", VoxelGrid))


@dataclass
class SceneCfg:
    """the obstacles of one world, by kind (reference ``SceneCfg``; ``curobo.scene.Scene``)"""

    sphere: Optional[List[Sphere]] = None
    cuboid: Optional[List[Cuboid]] = None
    capsule: Optional[List[Capsule]] = None
    cylinder: Optional[List[Cylinder]] = None
    mesh: Optional[List[Mesh]] = None
    voxel: Optional[List[VoxelGrid]] = None
    #: every obstacle, in the reference's order (spheres, cuboids, capsules, meshes, cylinders, voxel grids)
    objects: Optional[List[Obstacle]] = None

    def __post_init__(self):
        for kind, _ in _KINDS:
            if getattr(self, kind) is None:
                setattr(self, kind, [])
        if self.objects is None:
            self.objects = [o for kind, _ in _KINDS for o in getattr(self, kind)]

    def __len__(self) -> int:
        return len(self.objects)

    def __getitem__(self, idx: int) -> Obstacle:
        return self.objects[idx]

    def clone(self) -> "SceneCfg":
        return SceneCfg(**{kind: list(getattr(self, kind)) for kind, _ in _KINDS})

    @staticmethod
    def create(data_dict: Dict[str, Any]) -> "SceneCfg":
        """``{"cuboid": {name: {"dims": ..., "pose": ...}}, "sphere": {...}, "mesh": {...}, ...}`` (the yaml files under
        ``content/configs/scene/``) -> SceneCfg"""
        kw = {}
        for kind, cls in _KINDS:
            if kind in data_dict and data_dict[kind]:
                known = set(cls.__dataclass_fields__)
                kw[kind] = [cls(name=name, **{k: v for k, v in copy.deepcopy(dict(c)).items() if k in known}) for name, c in data_dict[kind].items()]
        return SceneCfg(**kw)

    def get_cache_dict(self) -> Dict[str, int]:
        return {"obb": len(self.cuboid), "mesh": len(self.mesh)}

    def add_obstacle(self, obstacle: Obstacle) -> None:
        for kind, cls in _KINDS:
            if type(obstacle) is cls:
                getattr(self, kind).append(obstacle)
                self.objects.append(obstacle)
                return
        raise ValueError(f"Obstacle type not supported: {type(obstacle).__name__}")

    def get_obstacle(self, name: str) -> Optional[Obstacle]:
        for o in self.objects:
            if o.name == name:
                return o
        return None

    def remove_obstacle(self, name: str) -> None:
        """drops the obstacle from ``objects`` AND from the list of its kind (the reference deletes it from ``objects`` only,
        geom/types.py:1281-1290, so that a scene rebuilt from the per-kind lists still contains it)"""
        o = self.get_obstacle(name)
        if o is None:
            return
        self.objects.remove(o)
        for kind, _ in _KINDS:
            lst = getattr(self, kind)
            if o in lst:
                lst.remove(o)

    @staticmethod
    def create_obb_world(current_world: "SceneCfg") -> "SceneCfg":
        """every analytic obstacle and mesh replaced by the oriented box around it (reference :1028-1052)"""
        boxes = [o.get_cuboid() for kind in ("sphere", "capsule", "cylinder", "mesh", "cuboid") for o in getattr(current_world, kind)]
        return SceneCfg(cuboid=boxes)

    def get_obb_world(self) -> "SceneCfg":
        return SceneCfg.create_obb_world(self)

    # ---- this package's loaders
    def to_config(self) -> Dict[str, Dict[str, Dict]]:
        """the dictionary form ``SceneCfg.create`` reads (``voxel`` entries keep their tensors)"""
        out: Dict[str, Dict[str, Dict]] = {}
        for kind, cls in _KINDS:
            for o in getattr(self, kind):
                d = {k: getattr(o, k) for k in cls.__dataclass_fields__ if k not in ("name", "position") and getattr(o, k) is not None}
                out.setdefault(kind, {})[o.name] = d
        return out


def as_scene_config(scene_model: Union["SceneCfg", Dict, str, None]):
    """SceneCfg -> its dictionary form; everything else passes through"""
    return scene_model.to_config() if isinstance(scene_model, SceneCfg) else scene_model
