"""Scene descriptions in the reference's configuration format -> the obstacle stores of the kernels.

Counterpart of ``SceneCfg.create`` (reference ``curobo/_src/geom/types.py:919-1010``; yaml files under
``content/configs/scene/``): a dictionary ``{"cuboid": {name: {"dims": [x, y, z], "pose": [x, y, z, qw, qx, qy, qz]}},
"sphere": {name: {"radius": r, "pose": [...]}}, "capsule": {name: {"radius", "base", "tip", "pose"}}, "cylinder":
{name: {"radius", "height", "pose"}}, "mesh": {...}, "voxel": {...}}``, a path to such a yaml file, a ``SceneCfg`` (scene/types.py:
``curobo.scene.Scene`` built from ``Cuboid`` / ``Sphere`` / ... objects), or a list of such (one per environment)."""

from __future__ import annotations

import os
from typing import Dict, List, Optional, Union

import numpy as np

from .data import cuboid_scene_arrays

#: the scene the reference's benchmarks name by file (content/configs/scene/collision_table.yml: one table cuboid)
PACKAGED_SCENES = {
    "collision_table.yml": {"cuboid": {"table": {"dims": [4.0, 4.0, 0.2], "pose": [0.0, 0.0, -0.2, 1, 0, 0, 0.0]}}},
}


def _one_env(cfg: Dict) -> List[Dict]:
    obs = []
    for name, c in (cfg.get("cuboid") or {}).items():
        obs.append({"dims": list(c["dims"]), "pose": list(c["pose"]), "enable": c.get("enable", True), "name": name})
    for name, c in (cfg.get("sphere") or {}).items():
        pose = list(c["pose"]) if c.get("pose") is not None else list(c["position"]) + [1, 0, 0, 0]
        obs.append({"type": "sphere", "radius": float(c["radius"]), "pose": pose, "enable": c.get("enable", True), "name": name})
    for name, c in (cfg.get("capsule") or {}).items():
        obs.append({"type": "capsule", "radius": float(c["radius"]), "base": list(c.get("base", [0, 0, 0])),
                    "tip": list(c.get("tip", [0, 0, 0])), "pose": list(c["pose"]), "enable": c.get("enable", True), "name": name})
    for name, c in (cfg.get("cylinder") or {}).items():
        obs.append({"type": "cylinder", "radius": float(c["radius"]), "height": float(c["height"]), "pose": list(c["pose"]),
                    "enable": c.get("enable", True), "name": name})
    return obs


def _plain(scene_model):
    """``SceneCfg`` objects (scene/types.py; one, or a list with one per environment) -> their dictionary form"""
    from .types import SceneCfg

    if isinstance(scene_model, SceneCfg):
        return scene_model.to_config()
    if isinstance(scene_model, (list, tuple)):
        return [m.to_config() if isinstance(m, SceneCfg) else m for m in scene_model]
    return scene_model


def _load_dicts(scene_model) -> List[Optional[Dict]]:
    """per environment the scene dictionary (yaml files read), ``None`` where the entry is already an obstacle list"""
    models = scene_model if isinstance(scene_model, (list, tuple)) else [scene_model]
    out = []
    for m in models:
        if isinstance(m, str):
            if m in PACKAGED_SCENES:
                m = PACKAGED_SCENES[m]
            else:
                import yaml

                with open(m) as fh:
                    m = yaml.safe_load(fh)
        out.append(m if isinstance(m, dict) else None)
    return out


def voxel_arrays_from_config(scene_model) -> Optional[Dict[str, np.ndarray]]:
    """the ``voxel`` entries (reference ``VoxelGrid``: ``dims``, ``voxel_size``, ``feature_tensor`` = the ESDF, ``pose``) of
    every environment -> the voxel arrays of ``SceneData.from_arrays``: fp16 features padded to the largest grid"""
    from .data import inverse_pose7

    scene_model = _plain(scene_model)
    if scene_model is None:
        return None
    envs = [list((m.get("voxel") or {}).items()) if m is not None else [] for m in _load_dicts(scene_model)]
    n = max(len(e) for e in envs)
    if n == 0:
        return None
    E = len(envs)
    shape = lambda c: [int(round(float(x) / float(c.get("voxel_size", 0.02)))) for x in c["dims"]]  # noqa: E731
    cells = max(int(np.prod(shape(c))) for e in envs for _, c in e)
    params, inv = np.zeros((E, n, 4), np.float32), np.zeros((E, n, 8), np.float32)
    inv[..., 3] = 1.0
    enable, count = np.zeros((E, n), np.uint8), np.zeros((E,), np.int32)
    feats = np.full((E, n, cells), -65504.0, np.float16)
    for e, grids in enumerate(envs):
        count[e] = len(grids)
        for g, (name, c) in enumerate(grids):
            nx, ny, nz = shape(c)
            f = c.get("feature_tensor")
            if f is None:
                raise ValueError(f"voxel grid '{name}' has no feature_tensor (the ESDF)")
            f = f.detach().cpu().numpy() if hasattr(f, "detach") else np.asarray(f)
            if f.size != nx * ny * nz:
                raise ValueError(f"voxel grid '{name}': feature_tensor has {f.size} values, dims / voxel_size give {nx} x {ny} x {nz}")
            params[e, g] = [nx, ny, nz, float(c.get("voxel_size", 0.02))]
            inv[e, g, :7] = inverse_pose7(c.get("pose") or [0, 0, 0, 1, 0, 0, 0])
            enable[e, g] = 1 if c.get("enable", True) else 0
            feats[e, g, : nx * ny * nz] = f.reshape(-1).astype(np.float16)
    names = [[name for name, _ in grids] + [None] * (n - len(grids)) for grids in envs]
    return {"voxel_params": params, "voxel_inv_pose": inv, "voxel_enable": enable, "voxel_count": count, "voxel_features": feats,
            "voxel_names": names}


def mesh_envs_from_config(scene_model) -> Optional[List[List[Dict]]]:
    """the ``mesh`` entries of a scene description, per environment, in the form ``scene.mesh.MeshStore`` takes: ``{"mesh":
    {name: {"file_path": *.obj | "vertices": ..., "faces": ..., "pose": [...], "scale": [...]}}}`` (reference ``Mesh``,
    geom/types.py); ``None`` when no environment has one"""
    scene_model = _plain(scene_model)
    if scene_model is None:
        return None
    envs = [[dict(c, name=name) for name, c in ((m.get("mesh") if m is not None else None) or {}).items()] for m in _load_dicts(scene_model)]
    return envs if any(envs) else None


def load_scene_config(scene_model: Union[str, Dict, List, None]) -> Optional[List[List[Dict]]]:
    """-> obstacle lists per environment (input of ``cuboid_scene_arrays``), or ``None`` for no world"""
    scene_model = _plain(scene_model)
    if scene_model is None:
        return None
    if isinstance(scene_model, (list, tuple)):
        return [_one_env(load_scene_config(m)[0] if isinstance(m, str) else m) if not isinstance(m, list) else m for m in scene_model]
    if isinstance(scene_model, str):
        if scene_model in PACKAGED_SCENES:
            return [_one_env(PACKAGED_SCENES[scene_model])]
        if not os.path.exists(scene_model):
            raise FileNotFoundError(f"scene configuration {scene_model!r} (packaged: {sorted(PACKAGED_SCENES)})")
        import yaml

        with open(scene_model) as fh:
            return [_one_env(yaml.safe_load(fh))]
    return [_one_env(scene_model)]


def scene_arrays_from_config(scene_model, max_n: Optional[int] = None) -> Optional[Dict[str, np.ndarray]]:
    envs = load_scene_config(scene_model)
    return None if envs is None else cuboid_scene_arrays(envs, max_n=max_n)


def _scene_cfgs(scene_model):
    """the description as ``SceneCfg`` objects (one, or a list with one per environment); ``None`` when it is raw obstacle lists"""
    from .types import SceneCfg

    def one(m, d):
        return m if isinstance(m, SceneCfg) else (SceneCfg.create(d) if isinstance(d, dict) else None)

    dicts = _load_dicts(_plain(scene_model))
    if isinstance(scene_model, (list, tuple)):
        out = [one(m, d) for m, d in zip(scene_model, dicts)]
        return None if any(o is None for o in out) else out
    return one(scene_model, dicts[0])


def scene_from_config(scene_model, device, gradient_mode: Optional[int] = None, cache: Optional[Dict[str, int]] = None):
    """scene description -> ``SceneData`` on ``device`` with every obstacle kind it names (cuboids and analytic primitives in the
    cuboid store, ``mesh`` entries behind their BVHs); ``None`` for no world.  What ``SceneCollision.from_config`` does with a
    ``SceneCfg`` in the reference (geom/collision/collision_scene.py).

    ``gradient_mode`` of the mesh store: ``None`` = ``MeshStore.CONSISTENT_GRADIENT`` (``CUROBO_MESH_GRADIENT=reference`` for
    the other).  A STATED DEVIATION of the scenes the solvers build: the reference's mesh query (data_mesh.py:693-697) returns
    (p - closest) / |p - closest| on both sides of the surface, which for a sphere centre OUTSIDE the mesh is the opposite of
    what its cuboid and voxel queries hand to the same kernel -- the cost then pulls every sphere that touches a mesh inward
    until its centre sits on the surface, and no optimiser leaves that state (tools/r06/mesh_vs_cuboid_plan.py: 0 of 12 seeds
    around a pillar as a mesh against 3 of 12 as a cuboid; the same 3 of 12 with the consistent vector).  The launch itself keeps the
    reference's vector as mode 0 (``MeshStore(...)`` default), which is what the parity tests hold against the oracle."""
    from .data import SceneData
    from .mesh import MeshStore

    if gradient_mode is None:
        gradient_mode = {"reference": MeshStore.REFERENCE_GRADIENT, "consistent": MeshStore.CONSISTENT_GRADIENT}[
            os.environ.get("CUROBO_MESH_GRADIENT", "consistent")]

    given, scene_model = scene_model, _plain(scene_model)
    cache = cache or {}
    # ``cache`` (reference SceneCollisionCfg.cache): slots to reserve per kind, so that ``add_obstacle`` has room later
    arrays = scene_arrays_from_config(scene_model, max_n=cache.get("cuboid", cache.get("obb")))
    if arrays is None:
        return None
    voxels = voxel_arrays_from_config(scene_model)
    if voxels is not None:
        arrays = dict(arrays, **voxels)
    meshes = mesh_envs_from_config(scene_model)
    store = MeshStore(meshes, device, gradient_mode=gradient_mode) if meshes is not None else None
    data = SceneData.from_arrays(arrays, device, meshes=store)
    try:
        data.scene_model = _scene_cfgs(given)
    except (TypeError, ValueError, OSError):  # (a description ``SceneCfg.create`` does not take: lookups by name stay with the stores)
        data.scene_model = None
    return data
