from .data import SceneData, cuboid_scene_arrays, inverse_pose7, validate_env_query_idx, voxel_grid_from_sdf  # noqa: F401
from .primitives import (bake_esdf, bake_mesh_esdf_device, box_mesh, capsule_sdf, cuboid_sdf, cylinder_sdf, mesh_sdf, sphere_sdf,  # noqa: F401
                         union_sdf)
from .mesh import MeshStore, load_mesh_file, load_obj, load_stl  # noqa: F401,E402
