from .data import SceneData, cuboid_scene_arrays, inverse_pose7, voxel_grid_from_sdf  # noqa: F401
