"""world description -> obstacle stores (``curobo_amd.scene``; reference curobo/scene.py: ``Scene`` and the obstacle types)"""
from curobo_amd.scene import MeshStore, SceneData, bake_mesh_esdf_device, cuboid_scene_arrays, load_obj, voxel_grid_from_sdf  # noqa: F401
from curobo_amd.scene.config import load_scene_config, scene_arrays_from_config, scene_from_config  # noqa: F401
from curobo_amd.scene.types import Capsule, Cuboid, Cylinder, Mesh, Obstacle, Sphere, VoxelGrid  # noqa: F401
from curobo_amd.scene.types import SceneCfg as Scene  # noqa: F401

__all__ = ["Scene", "SceneData", "Obstacle", "Cuboid", "Sphere", "Capsule", "Cylinder", "Mesh", "VoxelGrid",
           "cuboid_scene_arrays", "voxel_grid_from_sdf", "bake_mesh_esdf_device", "load_scene_config", "scene_arrays_from_config",
           "scene_from_config", "MeshStore", "load_obj"]
