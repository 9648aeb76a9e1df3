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
    if cfg.get("mesh"):
        raise ValueError("mesh obstacles: bake them into an ESDF grid with curobo_amd.scene.bake_mesh_esdf_device")
    return obs


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
