"""world description -> obstacle stores (``curobo_amd.scene``)"""
from curobo_amd.scene import MeshStore, SceneData, bake_mesh_esdf_device, cuboid_scene_arrays, load_obj, voxel_grid_from_sdf  # noqa: F401
from curobo_amd.scene.config import load_scene_config, scene_arrays_from_config, scene_from_config  # noqa: F401

__all__ = ["SceneData", "cuboid_scene_arrays", "voxel_grid_from_sdf", "bake_mesh_esdf_device", "load_scene_config", "scene_arrays_from_config", "scene_from_config", "MeshStore", "load_obj"]
