"""Scene descriptions in the reference's configuration format -> the obstacle stores of the kernels.

Counterpart of ``SceneCfg.create`` (reference ``curobo/_src/geom/types.py:919-1010``; yaml files under
``content/configs/scene/``): a dictionary ``{"cuboid": {name: {"dims": [x, y, z], "pose": [x, y, z, qw, qx, qy, qz]}},
"sphere": {name: {"radius": r, "pose": [...]}}, "capsule": {name: {"radius", "base", "tip", "pose"}}, "cylinder":
{name: {"radius", "height", "pose"}}}``, a path to such a yaml file, or a list of such (one per environment)."""

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


def mesh_envs_from_config(scene_model) -> Optional[List[List[Dict]]]:
    """the ``mesh`` entries of a scene description, per environment, in the form ``scene.mesh.MeshStore`` takes: ``{"mesh":
    {name: {"file_path": *.obj | "vertices": ..., "faces": ..., "pose": [...], "scale": [...]}}}`` (reference ``Mesh``,
    geom/types.py); ``None`` when no environment has one"""
    if scene_model is None:
        return None
    models = scene_model if isinstance(scene_model, (list, tuple)) else [scene_model]
    envs = []
    for m in models:
        if isinstance(m, str):
            if m in PACKAGED_SCENES:
                m = PACKAGED_SCENES[m]
            else:
                import yaml

                with open(m) as fh:
                    m = yaml.safe_load(fh)
        envs.append([dict(c, name=name) for name, c in ((m.get("mesh") if isinstance(m, dict) else None) or {}).items()])
    return envs if any(envs) else None


def load_scene_config(scene_model: Union[str, Dict, List, None]) -> Optional[List[List[Dict]]]:
    """-> obstacle lists per environment (input of ``cuboid_scene_arrays``), or ``None`` for no world"""
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


def scene_arrays_from_config(scene_model) -> Optional[Dict[str, np.ndarray]]:
    envs = load_scene_config(scene_model)
    return None if envs is None else cuboid_scene_arrays(envs)


def scene_from_config(scene_model, device, gradient_mode: int = 0):
    """scene description -> ``SceneData`` on ``device`` with every obstacle kind it names (cuboids and analytic primitives in the
    cuboid store, ``mesh`` entries behind their BVHs); ``None`` for no world.  What ``SceneCollision.from_config`` does with a
    ``SceneCfg`` in the reference (geom/collision/collision_scene.py)."""
    from .data import SceneData
    from .mesh import MeshStore

    arrays = scene_arrays_from_config(scene_model)
    if arrays is None:
        return None
    meshes = mesh_envs_from_config(scene_model)
    store = MeshStore(meshes, device, gradient_mode=gradient_mode) if meshes is not None else None
    return SceneData.from_arrays(arrays, device, meshes=store)
