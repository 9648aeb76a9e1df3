/*
 * curobo_hip.h -- C ABI of libcurobo_hip.so, the MI355X (gfx950) kernel backend for cuRobo's
 * batched motion-generation hot path.
 *
 * This is the drop-in boundary: every entry point replaces one module-level launch function of
 * the reference's kernel backend (curobo/_src/curobolib/backends/__init__.py:162-227; the two
 * existing backends are cuda_core_backend/ (Python) and pybind/ (C++ bindings)).  Argument order and
 * meaning follow the reference function that each entry point cites; torch.Tensor arguments
 * become raw device pointers, and one trailing `stream` (hipStream_t, may be NULL = default
 * stream) replaces the reference's implicit torch.cuda.current_stream() lookup
 * (cuda_core_backend/kinematics.py:50-51).
 *
 * Contract (same as the reference):
 *   - the caller owns all memory; every output is pre-allocated and mutated in place;
 *   - no allocation, no host synchronisation and no host read of device data happens inside a
 *     launch, so every entry point is hipGraph-capturable;
 *   - tensors are contiguous, fp32 unless stated; index tables int16 / int8 / int32 / uint8;
 *   - return value: 0 on success, non-zero on error; curobo_hip_last_error() returns a
 *     thread-local message (the reference raises via log_and_raise; the Python shim in
 *     curobo_amd/backends converts the status to the same exception types).
 */
#ifndef CUROBO_HIP_H
#define CUROBO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *curobo_hip_stream_t; /* hipStream_t */

#define CUROBO_HIP_OK 0
#define CUROBO_HIP_ERR_INVALID 1 /* bad argument (reference: ValueError / RuntimeError) */
#define CUROBO_HIP_ERR_LAUNCH 2  /* hipGetLastError() != hipSuccess after a launch */

const char *curobo_hip_last_error(void);
/* ABI version of this header; bumped on any signature change.  2 = round 2 (new arguments on the extension entry
 * points: fused rollouts, seed IK, dense self collision, the L-BFGS iteration tail; the reference-shaped entry points are
 * unchanged). */
int curobo_hip_abi_version(void);
/* when non-zero every launch is followed by hipStreamSynchronize + error check
 * (reference runtime.debug, cuda_core_backend/launch_helper.py:13-19). */
void curobo_hip_set_debug_sync(int enabled);

/* ---------------------------------------------------------------- kinematics
 * reference: cuda_core_backend/kinematics.py:21-379, pybind/kinematics_bindings.cpp:128-237
 * kernels:   kernels/kinematics/kinematics_forward_kernel.cuh:20-433,
 *            kernels/kinematics/kinematics_backward_kernel.cuh:27-157
 * batch_size is the number of points N = batch * horizon (cuda_ops/kinematics.py:113).
 */
/* replaces launch_kinematics_forward (cuda_core_backend/kinematics.py:21-88) */
int curobo_hip_launch_kinematics_forward(
    float *link_pos, float *link_quat, float *batch_center_of_mass, float *global_cumul_mat,
    const float *joint_vec, const float *fixed_transform, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map,
    const int16_t *tool_frame_map, const float *joint_offset_map, int batch_size, int horizon,
    int n_joints, int num_links, int n_tool_frames, int compute_com, curobo_hip_stream_t stream);

/* replaces launch_kinematics_forward_spheres (cuda_core_backend/kinematics.py:91-190) */
int curobo_hip_launch_kinematics_forward_spheres(
    float *link_pos, float *link_quat, float *batch_robot_spheres, float *batch_center_of_mass,
    float *global_cumul_mat, const float *joint_vec, const float *fixed_transform,
    const float *robot_spheres, const float *link_masses_com, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const float *joint_offset_map, const int32_t *env_query_idx,
    int num_envs, int batch_size, int horizon, int n_joints, int num_spheres, int num_links,
    int n_tool_frames, int write_global_cumul, int compute_com, curobo_hip_stream_t stream);

/* replaces launch_kinematics_forward_spheres_jacobian (cuda_core_backend/kinematics.py:193-300) */
int curobo_hip_launch_kinematics_forward_spheres_jacobian(
    float *link_pos, float *link_quat, float *batch_robot_spheres, float *batch_center_of_mass,
    float *batch_jacobian, float *global_cumul_mat, const float *joint_vec,
    const float *fixed_transform, const float *robot_spheres, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map,
    const int16_t *tool_frame_map, const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const int32_t *env_query_idx, int num_envs, int batch_size,
    int horizon, int n_joints, int num_spheres, int num_links, int n_tool_frames,
    int write_global_cumul, int compute_com, curobo_hip_stream_t stream);

/* replaces launch_kinematics_backward (cuda_core_backend/kinematics.py:303-379).
 * grad_spheres_b (extension, may be NULL): a second sphere-gradient buffer that is added to
 * grad_spheres on the fly, so the self-collision and scene-collision gradient buffers can be
 * consumed without a separate elementwise add.  link_chain_len = number of entries of
 * link_chain_data (link_chain_data.shape[0] in the reference).  compute_jacobian_grad != 0 is rejected
 * (CUROBO_HIP_ERR_INVALID): the dJ/dq term is a SURVEY section 8f-2 "next" row. */
int curobo_hip_launch_kinematics_backward(
    float *grad_out, const float *grad_nlinks_pos, const float *grad_nlinks_quat,
    const float *grad_spheres, const float *grad_spheres_b, const float *grad_center_of_mass,
    const float *batch_center_of_mass, const float *grad_jacobian, const float *global_cumul_mat,
    const float *robot_spheres, const float *link_masses_com, const int16_t *link_map,
    const int16_t *joint_map, const int8_t *joint_map_type, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const int32_t *env_query_idx, int num_envs, int batch_size,
    int horizon, int n_joints, int num_spheres, int num_links, int n_tool_frames, int link_chain_len,
    int compute_com, int compute_jacobian_grad, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- geometry: self collision
 * reference: cuda_core_backend/geometry.py:63-227, pybind/geometry_bindings.cpp:16-45
 * kernels:   kernels/geometry/self_collision/self_collision_kernel.cuh:19-297
 * num_blocks_per_batch / max_threads_per_block are accepted for signature parity; the HIP
 * backend picks its own wave64 tiling (block_batch_max_* scratch is used when it splits the
 * pair list across workgroups).
 */
int curobo_hip_self_collision_distance(
    float *out_distance, float *out_vec, float *pair_distance, uint8_t *sparse_index,
    const float *robot_spheres, const float *sphere_padding, const float *weight,
    const int16_t *pair_locations, float *block_batch_max_value, int16_t *block_batch_max_index,
    int num_blocks_per_batch, int max_threads_per_block, int batch_size, int horizon, int nspheres,
    int num_collision_pairs, int store_pair_distance, int compute_grad,
    curobo_hip_stream_t stream);

/* Dense pair sets (humanoids: most of the S (S - 1) / 2 sphere pairs are enabled): the same result as
 * curobo_hip_self_collision_distance without store_pair_distance, from a register-tiled all-pairs pass masked by a
 * bitmap of the pair set instead of a gather per listed pair.  pair_bitmap: uint32 [2 * nslots][nslots * 64], bit jj
 * of pair_bitmap[jb][i] <-> pair (i, 32 * jb + jj) of pair_locations (i < j; the list must be (i, j)-sorted so
 * that "lowest pair index" = lexicographically first pair); nslots = multiple of 4 with nslots * 64 >= nspheres.
 * reference: self_collision_max_block_kernel + _max_reduce_kernel, self_collision_kernel.cuh:113-297.
 * tile_list (optional, NULL = evaluate every tile): the 16 x 16 tiles of the pair matrix that hold an enabled pair, as
 * (i / 16) | (j / 16) << 8; with it a tile is only evaluated when the bounding boxes of its two 16-sphere blocks overlap
 * (result preserving: only positive penetrations count).
 * With a tile list the narrow phase runs on the matrix cores: one v_mfma_f32_16x16x4_f32 per surviving tile gives all 256
 * penetrations of the tile in the expanded form 2 c_i.c_j + 2 r_i r_j + (r_i^2 - |c_i|^2) + (r_j^2 - |c_j|^2); that value
 * only culls, the listed pairs it leaves are evaluated again exactly (same bits as the other kernels).
 * tile_lane_masks (ABI 6, optional, NULL = read the pair bitmap instead): uint8 [num_tiles][64], bit reg of byte (c, lane)
 * <-> pair (16 ib + 4 (lane / 16) + reg, 16 jb + lane % 16) of tile c is listed: the tile's pair set in the layout in which
 * a lane receives the matrix-core result, 64 bytes per tile instead of 1 KB of bitmap words. */
int curobo_hip_self_collision_distance_dense(
    float *out_distance, float *out_vec, uint8_t *sparse_index, const float *robot_spheres,
    const float *sphere_padding, const float *weight, const uint32_t *pair_bitmap, const int32_t *tile_list, int num_tiles,
    const uint8_t *tile_lane_masks, int batch_size, int horizon, int nspheres, int nslots, int compute_grad,
    curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- collision: sphere vs scene
 * The reference has NO backend hook here: these are NVIDIA Warp kernels launched from
 * geom/collision/wp_autograd.py:37-249 (SphereObstacleCollision / SweptSphereObstacleCollision).
 * The struct mirrors CuboidDataWarp (geom/data/data_cuboid.py:43-62) and VoxelDataWarp
 * (geom/data/data_voxel.py:684-702).  One launch handles every obstacle type, sums the
 * obstacles of a sphere in index order (deterministic; the reference uses float atomics) and
 * fully rewrites distance/gradient (no separate zero_() pass).
 */
typedef struct curobo_hip_scene {
  const float *cuboid_dims;       /* [num_envs, max_cuboids, 4] full extents */
  const float *cuboid_inv_pose;   /* [num_envs, max_cuboids, 8] x y z qw qx qy qz pad */
  const uint8_t *cuboid_enable;   /* [num_envs, max_cuboids] */
  const int32_t *cuboid_count;    /* [num_envs] */
  int32_t max_cuboids;
  const float *voxel_params;      /* [num_envs, max_voxel_grids, 4] nx ny nz voxel_size */
  const float *voxel_inv_pose;    /* [num_envs, max_voxel_grids, 8] */
  const uint8_t *voxel_enable;    /* [num_envs, max_voxel_grids] */
  const int32_t *voxel_count;     /* [num_envs] */
  const uint16_t *voxel_features; /* fp16 ESDF [num_envs, max_voxel_grids, n_voxels] */
  int32_t max_voxel_grids;
  int32_t voxel_n_voxels;
  float voxel_max_distance;
  /* optional culling aid built at scene upload (NULL = none; results are identical with and without it):
   * fp16 [num_envs, max_voxel_grids, voxel_n_coarse], cell (cx, cy, cz) of a grid = the MINIMUM of its ESDF over
   * the block of voxel_coarse_block^3 voxels starting at (cx, cy, cz) * voxel_coarse_block, dilated by
   * voxel_coarse_dilate voxels on every side; row-major over (ceil(nx / block), ceil(ny / block), ceil(nz / block)).
   * A sphere whose whole sweep stays (voxel_coarse_dilate - 1) voxels around its centre's voxel and whose radius +
   * activation distance is below that minimum cannot touch the grid: its 8-corner lookups are skipped. */
  const uint16_t *voxel_coarse_min;
  int32_t voxel_coarse_block;
  int32_t voxel_coarse_dilate;
  int32_t voxel_n_coarse;
  /* Analytic primitives (beyond the reference, which turns Sphere / Capsule / Cylinder obstacles into meshes,
   * geom/types.py:1104-1124): a record of the cuboid store is a primitive when cuboid_dims[..., 3] (the reference's
   * zero padding) holds a tag: 1 sphere (dims = radius, -, -), 2 capsule (radius, half length of the segment on the
   * local z axis, -), 3 cylinder (radius, half height along local z, -); pose / enable / count as for cuboids.  Set
   * cuboid_has_primitives when any tag is non-zero (selects the kernel instantiations that test the tag). */
  int32_t cuboid_has_primitives;
} curobo_hip_scene;

/* sweep_steps: 0 = SphereObstacleCollision, 3 = SweptSphereObstacleCollision (SWEEP_STEPS,
 * wp_sweep_collision_kernel.py:66).  enable_speed_metric applies wp_speed_metric.py:10-93 in
 * the same launch; speed_dt is the reference's 1-element device tensor (may be NULL if off). */
int curobo_hip_sphere_obstacle_collision(
    float *distance, float *gradient, const float *spheres, const curobo_hip_scene *scene,
    const float *weight, const float *activation_distance, const int32_t *env_query_idx,
    int batch_size, int horizon, int num_spheres, int use_multi_env, int sweep_steps,
    int enable_speed_metric, const float *speed_dt, curobo_hip_stream_t stream);

/* Mesh obstacles -> one more ESDF grid of the voxel store (scene-upload time, not the hot path).  The reference
 * queries meshes through NVIDIA Warp's BVH (geom/data/data_mesh.py:555-700, wp.mesh_query_point); here a closed,
 * consistently oriented triangle mesh is baked into an fp16 grid [nx, ny, nz] (voxel centres
 * (i + 0.5 - n / 2) * voxel_size in the grid frame, the layout of geom/data/data_voxel.py:42-95): exact
 * point-triangle distance, sign from the generalised winding number, clamped to +-max_distance.
 * vertices [n_vertices, 3] (mesh frame) and faces int32 [n_faces, 3] are device pointers; grid_to_mesh_3x4_host is a
 * HOST pointer to the row-major 3x4 transform grid frame -> mesh frame. */
int curobo_hip_mesh_esdf_bake(uint16_t *out_esdf_fp16, const float *vertices, const int32_t *faces, int n_vertices,
                              int n_faces, int nx, int ny, int nz, float voxel_size, float max_distance,
                              const float *grid_to_mesh_3x4_host, curobo_hip_stream_t stream);

/* Triangle-mesh obstacles queried directly (the reference: geom/data/data_mesh.py:555-700, wp.mesh_query_point per query
 * sphere through NVIDIA Warp's BVH).  A mesh is a linear BVH in heap layout: triangles sorted by the Morton code of their
 * centroids, `leaf_size` consecutive triangles per leaf, `n_leaves` (a power of two) leaves, node k has the children 2k and
 * 2k + 1, the leaves are the nodes n_leaves .. 2 n_leaves - 1.  tri: [n_tri][12] floats = (a, b - a, c - a) as float4 each,
 * in sorted order; node_box: [2 * n_leaves][8] floats = (lo xyz, -, hi xyz, -), node 0 unused.  All device pointers. */
typedef struct curobo_hip_mesh {
  const float *tri;
  const float *node_box;
  /* optional (ABI 5): [n_tri][6][4] floats in the sorted triangle order = the angle-weighted pseudonormals of the triangle's
   * vertices a, b, c and the edge pseudonormals of ab, bc, ca (sum of the normals of the faces that share the feature;
   * Baerentzen & Aanaes 2005).  The sign of a query whose closest point lies on an edge or a vertex is the sign of
   * (point - closest) . pseudonormal; NULL = count ray crossings instead (three tree walks per such query). */
  const float *tri_pn;
  int32_t n_tri, n_leaves, leaf_size;
  /* ABI 7 (was padding = 0).  How inside / outside is decided:
   *   0 = by the closest feature (face normal / pseudonormal; crossings counted when that gives no verdict): the same function
   *       as the reference's rule on a closed, consistently oriented surface, without a ray;
   *   1 = the reference's rule as published -- Warp's mesh_query_point -> mesh_query_inside (warp/native/mesh.h; called from
   *       data_mesh.py:632,682): rays from the query point along +x, +y, +z, inside iff all three hit and every ray's NEAREST
   *       hit is a back face.  The rule for meshes that are open or not consistently oriented, where the two differ. */
  int32_t sign_rule;
  /* ABI 7, optional: distance-sorted closest-triangle cell lists over a uniform grid in the mesh frame (NULL = every query
   * walks the tree).  The grid covers the bounding box grown by grid_pad, cells of edge grid_h, cell (ix, iy, iz) has the
   * index (ix * grid_n[1] + iy) * grid_n[2] + iz.  cell_start [n_cells + 1][2]: word 0: bits 0..29 = first entry of the cell's list in
   * cell_list, bits 30..31 = 1 / 2 when every point of the cell is outside / inside the surface (sign_rule 0 only), else 0;
   * word 1: the distance of the cell's centre from the surface (float bits).
   * cell_list: 16-byte entries (int32 triangle index in sorted order, float distance of that triangle from the CELL CENTRE c,
   * two floats = the octahedral code of the unit vector n from the triangle's closest point towards c), ascending in distance,
   * closed by a sentinel (-1, cover, 0, 0): every triangle not listed is farther than `cover` from the centre.  A query point p
   * at distance delta from its cell's centre goes through the list in order: an entry needs its triangle tested only when its
   * bound distance + (p - c) . n (the triangle lies behind the plane through its closest point with normal n) does not exceed
   * the best distance found; the query stops at the first entry whose distance - delta exceeds it (every later triangle is
   * farther still: the result is the exact closest point); a list that ends before that (best > cover - delta) sends the query
   * to the tree walk. */
  const uint32_t *cell_start;
  const int32_t *cell_list;
  float grid_lo[3], grid_h;
  int32_t grid_n[3];
  float grid_pad;
} curobo_hip_mesh;

/* The mesh obstacles of a scene (layout of the reference's MeshData, data_mesh.py:60-120): meshes = DEVICE array of
 * curobo_hip_mesh (the cache of loaded meshes), mesh_id [num_envs, max_n] picks one per obstacle slot, dims [num_envs,
 * max_n, 4] = bounding-box extents (the query's max_distance is half their diagonal), inv_pose / enable / count as for
 * cuboids.  gradient_mode 0: the local gradient exactly as data_mesh.py:693-697 computes it, (p - closest) / |p - closest|
 * on either side of the surface; 1: that vector negated for centres outside the surface, i.e. minus the gradient of the
 * signed distance everywhere, which is what the cuboid (data_cuboid.py:596-626) and voxel kinds hand to the same kernel. */
/* some mesh of the set carries cell lists: the queued launch answers what it can through them before it walks trees */
#define CUROBO_HIP_MESH_SET_HAS_CELLS 1
typedef struct curobo_hip_mesh_set {
  const curobo_hip_mesh *meshes;
  const int32_t *mesh_id;
  const float *dims;
  const float *inv_pose;
  const uint8_t *enable;
  const int32_t *count;
  int32_t max_n, gradient_mode;
  int32_t num_envs;        /* leading dimension of mesh_id / dims / inv_pose / enable / count (ABI 5) */
  int32_t flags;           /* ABI 7 (was padding = 0): CUROBO_HIP_MESH_SET_HAS_CELLS */
} curobo_hip_mesh_set;

/* Build: (1) Morton keys of the triangle centroids inside bounds_lo_hi_host (HOST pointer, 6 floats: the mesh's bounding
 * box) -> out_codes [n_faces] int64 = code << 32 | triangle index; (2) the caller sorts the keys (any sort: they are
 * unique); (3) triangles in sorted order + leaf boxes + one launch per level of the tree for the inner boxes.  out_tri
 * [n_faces * 12], out_node_box [2 * n_leaves * 8] floats. */
int curobo_hip_mesh_morton_codes(int64_t *out_codes, const float *vertices, const int32_t *faces, int n_faces,
                                 const float *bounds_lo_hi_host, curobo_hip_stream_t stream);
int curobo_hip_mesh_bvh_build(float *out_tri, float *out_node_box, const float *vertices, const int32_t *faces,
                              const int64_t *sorted_codes, int n_faces, int n_leaves, int leaf_size,
                              curobo_hip_stream_t stream);

/* The cell lists of a mesh (curobo_hip_mesh.cell_start / cell_list), in two launches around the caller's prefix sum and sort
 * (plumbing, as for the Morton keys).  `mesh` (HOST pointer) carries the tree and the grid fields; its cell pointers are not
 * read.  (1) count: per cell the length of its list including the sentinel -> out_count [n_cells], its cover radius ->
 * out_cover [n_cells] (0 = the cell has no list: more than gather_cap triangles about equally far), the side of the whole
 * cell -> out_side [n_cells] (1 outside / 2 inside / 0 may straddle), the distance of its centre from the surface ->
 * out_centre_dist [n_cells].  (2) fill, given offsets [n_cells + 1] = the exclusive
 * prefix sum of out_count as int64: out_entries [offsets[n_cells]][4] = (triangle, distance bits, direction code x, y), unsorted within a cell,
 * out_keys [offsets[n_cells]] = cell << 32 | distance bits (sentinel: 0x7fffffff), out_cell_start [n_cells + 1][2] = the packed
 * words of curobo_hip_mesh.cell_start.  The caller sorts the keys and gathers the entries with the permutation. */
int curobo_hip_mesh_cells_count(int32_t *out_count, float *out_cover, uint8_t *out_side, float *out_centre_dist,
                                const curobo_hip_mesh *mesh, int gather_cap, curobo_hip_stream_t stream);
int curobo_hip_mesh_cells_fill(int64_t *out_keys, int32_t *out_entries, uint32_t *out_cell_start, const int64_t *offsets,
                               const float *cover, const uint8_t *side, const float *centre_dist,
                               const curobo_hip_mesh *mesh, curobo_hip_stream_t stream);

/* compute_local_sdf_with_grad of data_mesh.py:630-700 for points [n, 3] in the mesh frame: out_sdf [n] = signed distance
 * (negative inside; max_distance when no surface lies within max_distance), out_grad [n, 3] (may be NULL) = (point -
 * closest point) / distance.  Exact point-triangle distances; the sign is the parity of ray crossings (closed meshes). */
int curobo_hip_mesh_query(float *out_sdf, float *out_grad, const float *points, const curobo_hip_mesh *mesh,
                          float max_distance, int n_points, curobo_hip_stream_t stream);

/* curobo_hip_mesh_esdf_bake through the BVH: O(voxels x log triangles) instead of O(voxels x triangles). */
int curobo_hip_mesh_esdf_bake_bvh(uint16_t *out_esdf_fp16, const curobo_hip_mesh *mesh, int nx, int ny, int nz,
                                  float voxel_size, float max_distance, const float *grid_to_mesh_3x4_host,
                                  curobo_hip_stream_t stream);

/* Sphere-vs-mesh collision: the mesh share of SphereObstacleCollision / SweptSphereObstacleCollision.forward
 * (geom/collision/wp_autograd.py:37-249 launches its kernel once per obstacle kind into the same buffers; this is the
 * launch for the mesh kind).  accumulate != 0: add to distance / gradient as written by curobo_hip_sphere_obstacle_collision
 * for the other kinds (the speed metric is applied to this share: it is linear); 0: overwrite them. */
int curobo_hip_sphere_mesh_collision(
    float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres,
    int use_multi_env, int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate,
    curobo_hip_stream_t stream);

/* The same launch with a caller-owned device workspace (>= curobo_hip_sphere_mesh_collision_ws_bytes(...) bytes, 16-byte
 * aligned, contents irrelevant; the size is written to the HOST pointer out_bytes_host): the spheres that survive the bounding-box reject of any mesh -- few, and clustered in the
 * batch -- are queued launch-wide and their tree walks run on a grid that covers the chip once, instead of inside the
 * workgroups that happen to hold them.  Same results (per sphere the slots are summed in ascending order in both forms);
 * kernels only on `stream` (since the end of round 6 the queue counters -- the first 16 bytes: spheres queued from the head,
 * handed to the tree walk, queued from the tail, handed to a workgroup of their own -- are cleared by a kernel: a captured
 * 16-byte memset node faulted at the second replay of its graph), graph-capturable: counter reset, select, and with cell lists
 * the cell-list kernel, the workgroup-per-sphere kernel and the tree walk of what is left. */
int curobo_hip_sphere_mesh_collision_ws_bytes(int batch_size, int horizon, int num_spheres, int64_t *out_bytes_host);
int curobo_hip_sphere_mesh_collision_ws(
    float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres,
    int use_multi_env, int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate, void *workspace,
    size_t workspace_bytes, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- cost: tool pose + c-space
 * The reference runs these as NVIDIA Warp kernels without a backend hook:
 * ToolPoseDistance (cost/wp_tool_pose.py:698-914, kernel :456-692) and the POSITION c-space cost
 * (cost/wp_cspace_position.py:232-362).  Argument order follows the Warp kernels' inputs.
 * goal_position/goal_quat: [n_goals, num_links, num_goalset, 3|4] (quaternions wxyz);
 * out_distance [b,h,2*num_links] = (position cost, rotation cost) per link; out_rotation_gradient
 * is the quaternion rate q (x) (omega,0) (wxyz) that launch_kinematics_backward consumes.
 * rotation_method: 0 axis-angle, 1 lie group, 2 lie group advanced. */
int curobo_hip_tool_pose_distance(
    float *out_distance, float *out_position_distance, float *out_rotation_distance,
    float *out_position_gradient, float *out_rotation_gradient, int32_t *out_goalset_idx,
    const float *current_position, const float *current_quat, const float *goal_position,
    const float *goal_quat, const int32_t *idxs_goal, const float *position_orientation_weight,
    const float *terminal_pose_axes_weight_factor, const float *non_terminal_pose_axes_weight_factor,
    const float *terminal_pose_convergence_tolerance,
    const float *non_terminal_pose_convergence_tolerance, const uint8_t *project_distance_to_goal,
    int batch_size, int horizon, int num_links, int num_goalset, int rotation_method,
    curobo_hip_stream_t stream);

/* p_b/effort_b/v_b: [2, dof] lower then upper; weight/activation_distance: [2] (position,
 * effort); squared_l2_reg_weight: [2] (velocity, acceleration); effort / out_grad_tau may be NULL. */
int curobo_hip_cspace_position_cost(
    float *out_cost, float *out_grad_p, float *out_grad_tau, const float *pos, const float *effort,
    const float *cspace_target, const int32_t *cspace_target_idx, const float *p_b,
    const float *effort_b, const float *weight, const float *activation_distance,
    const float *cspace_target_weight, const float *cspace_target_dof_weight,
    const float *squared_l2_reg_weight, const float *current_position,
    const float *current_velocity, const int32_t *idxs_current_state, const float *v_b,
    const float *state_dt, int write_grad, int batch_size, int horizon, int dof,
    curobo_hip_stream_t stream);

/* c-space STATE cost (reference cost/wp_cspace_state.py:20-287, a Warp kernel): bound costs on
 * position / velocity / acceleration / jerk / effort, optional joint target, squared-L2 and
 * energy regularisation.  weight, activation_distance, squared_l2_regularization_weights: [5];
 * limits [2, dof] each; state_dt [batch]; effort / out_grad_tau may be NULL. */
int curobo_hip_cspace_state_cost(
    float *out_cost, float *out_grad_p, float *out_grad_v, float *out_grad_a, float *out_grad_j,
    float *out_grad_tau, const float *pos, const float *vel, const float *acc, const float *jerk,
    const float *effort, const float *state_dt, const float *target_joint_position,
    const int32_t *idxs_target_joint_position, const float *p_b, const float *v_b,
    const float *a_b, const float *j_b, const float *effort_b, const float *weight,
    const float *activation_distance, const float *squared_l2_regularization_weights,
    const float *cspace_target_weight, const float *cspace_non_terminal_weight_factor,
    const float *cspace_target_dof_weight, int write_grad, int batch_size, int horizon, int dof,
    int retime_weights, int retime_regularization_weights, curobo_hip_stream_t stream);

/* Per-row aggregation for horizon-1 (teleport / IK) rollouts (reference: torch cat+sum and autograd
 * accumulation, rollout/metrics.py:233-265): out_cost[r] = sum(pose_cost[r,:2*num_links]) +
 * sum(cspace_cost[r,:dof]) + self_cost[r] + sum(scene_cost[r,:num_spheres]);
 * grad_q[r,:] += cspace_grad[r,:].  Any input may be NULL. */
int curobo_hip_rollout_point_aggregate(
    float *out_cost, float *grad_q, const float *pose_cost, const float *cspace_cost,
    const float *cspace_grad, const float *self_cost, const float *scene_cost, int rows,
    int num_links, int dof, int num_spheres, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- dynamics: RNEA
 * reference: cuda_core_backend/dynamics.py:24-131,134-260
 * kernels:   kernels/dynamics/rnea_forward_kernel.cuh:53-292, rnea_backward_kernel.cuh:65-468
 * tau[b, num_dof] = RNEA(q, qd, qdd, f_ext); spatial vectors are [angular; linear], gravity[6] is
 * the spatial base acceleration (0,0,0,0,0,+9.81 for z-up gravity).  forward_cache is the
 * reference's opaque [batch, num_links*20] scratch handed from forward to backward (internal
 * layout here: [link][20][batch]).  level_starts / n_levels / threads_per_batch are accepted for
 * signature parity; links are visited in level_links order.  f_ext / grad_f_ext [b, links, 6]
 * may be NULL.  The backward needs a caller-owned workspace of num_links*18*batch floats (the
 * Python shim keeps one per device). */
int curobo_hip_launch_rnea_forward(
    float *tau, const float *q, const float *qd, const float *qdd, const float *fixed_transforms,
    const float *link_masses_com, const float *link_inertias, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const float *joint_offset_map,
    const float *gravity, const int16_t *level_starts, const int16_t *level_links,
    float *forward_cache, int batch_size, int num_links, int num_dof, int n_levels,
    int threads_per_batch, const float *f_ext, curobo_hip_stream_t stream);

int curobo_hip_launch_rnea_backward(
    float *grad_q, float *grad_qd, float *grad_qdd, const float *grad_tau, const float *q,
    const float *qd, const float *fixed_transforms, const float *link_masses_com,
    const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const float *joint_offset_map, const float *gravity,
    const int16_t *level_starts, const int16_t *level_links, const float *forward_cache,
    int batch_size, int num_links, int num_dof, int n_levels, int threads_per_batch,
    float *grad_f_ext, float *workspace, curobo_hip_stream_t stream);

/* Extensions: the same two launches with a caller-provided scratch of 3 * num_dof * batch_size floats.  The launch first
 * transposes its joint-space inputs (q, qd, qdd | grad_tau) into it, [dof][batch], and the walks read them from there:
 * coalesced without staging them through LDS (38 KB per workgroup for a 49-dof humanoid).  Same values as the launches
 * above; meant for rollouts that run the walks NEXT TO a kernel that lives on LDS (C4: the self-collision kernel holds
 * eight points per CU in 160 KB, and three while the staged walks share the CU).  flags bit 0: the scratch still holds q and qd of a
 * forward launch on the same inputs (its first two thirds): only grad_tau is transposed; bit 1: the gradients are ADDED to
 * grad_q / grad_qd / grad_qdd (the caller's running joint-space gradients) instead of overwriting them. */
int curobo_hip_launch_rnea_forward_scratch(
    float *tau, const float *q, const float *qd, const float *qdd, const float *fixed_transforms,
    const float *link_masses_com, const float *link_inertias, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const float *joint_offset_map,
    const float *gravity, const int16_t *level_starts, const int16_t *level_links,
    float *forward_cache, int batch_size, int num_links, int num_dof, int n_levels,
    int threads_per_batch, const float *f_ext, float *scratch, curobo_hip_stream_t stream);

int curobo_hip_launch_rnea_backward_scratch(
    float *grad_q, float *grad_qd, float *grad_qdd, const float *grad_tau, const float *q,
    const float *qd, const float *fixed_transforms, const float *link_masses_com,
    const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const float *joint_offset_map, const float *gravity,
    const int16_t *level_starts, const int16_t *level_links, const float *forward_cache,
    int batch_size, int num_links, int num_dof, int n_levels, int threads_per_batch,
    float *grad_f_ext, float *workspace, float *scratch, int flags, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- linalg: Levenberg-Marquardt step
 * reference: optim/util/levenberg_marquardt_step.py:96-199 (Warp tile kernel, no backend hook).
 * Per problem: delta = -(J^T J + lambda I)^-1 jTerror; joint_position_out = joint_position_in +
 * delta; pred_reduction = 0.5 * delta . (lambda * delta - jTerror).
 * jacobian [batch, n_residuals, action_dim], action_dim <= 64.  J^T J runs on the matrix cores
 * (v_mfma_f32_16x16x4_f32, exact fp32), the Cholesky solve in LDS, one wavefront per problem. */
int curobo_hip_levenberg_marquardt_step(
    float *joint_position_out, float *pred_reduction, const float *jacobian, const float *jTerror,
    const float *lambda_damping, const float *joint_position_in, int batch_size, int n_residuals,
    int action_dim, curobo_hip_stream_t stream);

/* c-space L2 distance cost (reference forward_l2_warp / L2DistFunction,
 * cost/wp_torch_cspace_dist.py:12-158): out_cost[b, h, d] = weight[0] * r[d] * (pos - target[target_idx[b]])^2
 * with r = terminal_dof_weight at h == horizon - 1, else non_terminal_dof_weight [dof]; out_grad_p =
 * 2 w err.  Entries of zero weight are left untouched, as in the reference. */
int curobo_hip_cspace_l2_distance(float *out_cost, float *out_grad_p, const float *pos, const float *target,
                                  const int32_t *target_idx, const float *weight,
                                  const float *terminal_dof_weight, const float *non_terminal_dof_weight,
                                  int write_grad, int batch_size, int horizon, int dof,
                                  curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- seed IK: iteration-state update
 * reference (torch elementwise ops, ~25 launches per iteration):
 *   solver/seed_ik/seed_ik_error_calculator.py:292-305,338-387,464-495 (pose-error reduction,
 *   joint-limit residual rows, combination) and solver/seed_ik/seed_iteration_state_manager.py:74-260
 *   (trust ratio rho = (old - new) / (pred + 1e-8), accept rho >= rho_min, lambda /= or *= factor
 *   clamped, candidate-or-current selection, convergence flags).
 * State [n, ...] is updated in place from the candidate evaluation: candidate_pose_jacobian
 * [n, 6T, dof] (FK Jacobian kernel), candidate_pose_jTerror [n, dof] (FK VJP of the pose cost),
 * candidate_pose_cost [n, T, 2], candidate_{position,rotation}_distance [n, T] (tool-pose kernel),
 * predicted_reduction [n] (LM step).  jacobian [n, 6T + dof, dof] gets the pose rows and the
 * diagonal joint-limit rows; error_norm always takes the candidate's value (as the reference does).
 * initial != 0: the candidate becomes the state unconditionally (lambda is left as set by the
 * caller).  current_position / dt / velocity_limits [2, dof] (optional, all or none) tighten the
 * limits for velocity-aware IK.  * velocity / acceleration regularisation rows (seed_ik_error_calculator.py:389-456): with current_position + dt given,
 * velocity_weight > 0 adds r_v = sqrt(w dt) (q - current_position) / dt and acceleration_weight > 0 (current_velocity
 * given) r_a = sqrt(w) ((q - current_position) / dt - current_velocity); velocity_limits may be NULL (no clamping).
 */
int curobo_hip_seed_ik_update_state(
    float *joint_position, float *jacobian, float *jTerror, float *error_norm, float *position_error,
    float *orientation_error, float *lambda_damping, uint8_t *success, uint8_t *improvement,
    const float *candidate_joint_position, const float *candidate_pose_jacobian,
    const float *candidate_pose_jTerror, const float *candidate_pose_cost,
    const float *candidate_position_distance, const float *candidate_rotation_distance,
    const float *predicted_reduction, const float *action_min, const float *action_max,
    const float *current_position, const float *dt, const float *velocity_limits,
    const float *current_velocity, float velocity_weight, float acceleration_weight,
    float joint_limit_weight, float rho_min, float lambda_factor, float lambda_min, float lambda_max,
    float convergence_position_tolerance, float convergence_orientation_tolerance,
    float convergence_joint_limit_weight, int num_problems, int dof, int num_tool_frames,
    int initial, curobo_hip_stream_t stream);

/* `iterations` whole Levenberg-Marquardt iterations of the seed-IK solver (LM step -> FK + tool-frame Jacobian ->
 * tool-pose error -> J^T e -> curobo_hip_seed_ik_update_state) in ONE launch, preceded by the initial evaluation of
 * seed_joint_position when `initial` != 0: a problem lives on one 16-lane row with its state in LDS, global memory sees
 * the state (the buffers of curobo_hip_seed_ik_update_state) on the way in and on the way out.  Replaces
 * 5 x iterations launches of curobo_hip_levenberg_marquardt_step, curobo_hip_launch_kinematics_forward_spheres_jacobian,
 * curobo_hip_tool_pose_distance, curobo_hip_launch_kinematics_backward, curobo_hip_seed_ik_update_state (reference:
 * solver/seed_ik/seed_ik_solver.py:48-824, one torch graph per inner loop).  dof <= 16.  Goal / weight pointers as in
 * curobo_hip_tool_pose_distance (horizon 1: terminal weights and tolerances). */
int curobo_hip_seed_ik_iterate(
    float *joint_position, float *jacobian, float *jTerror, float *error_norm, float *position_error,
    float *orientation_error, float *lambda_damping, uint8_t *success, uint8_t *improvement, const float *seed_joint_position,
    const float *goal_position, const float *goal_quat, const int32_t *idxs_goal, const float *position_orientation_weight,
    const float *pose_axes_weight_factor, const float *pose_convergence_tolerance, const uint8_t *project_distance_to_goal,
    int num_goalset, int rotation_method, const float *fixed_transform, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_chain_data, const int16_t *link_chain_offsets,
    const int16_t *joint_links_data, const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const float *action_min, const float *action_max, const float *current_position,
    const float *dt, const float *velocity_limits, const float *current_velocity, float velocity_weight,
    float acceleration_weight, float joint_limit_weight, float rho_min, float lambda_factor, float lambda_min,
    float lambda_max, float convergence_position_tolerance, float convergence_orientation_tolerance,
    float convergence_joint_limit_weight, int num_problems, int dof, int num_links, int num_tool_frames, int link_chain_len,
    int iterations, int initial, const int32_t *stop_flag, int32_t *blocks_run, curobo_hip_stream_t stream);

/* 1 when the per-problem state of curobo_hip_seed_ik_iterate (16 problems per workgroup: accepted / candidate Jacobian,
 * normal matrix, the candidate's link transforms, pose scratch) fits the launch's 64 KB of LDS and dof <= 16, else 0: the
 * caller then runs the five-launch iteration (curobo_hip_levenberg_marquardt_step ... curobo_hip_seed_ik_update_state).
 * Returns the answer itself, not a status (like curobo_hip_rollout_trajopt_fused_torque_fits). */
int curobo_hip_seed_ik_iterate_fits(int dof, int num_links, int num_tool_frames, int link_chain_len);

/* Device-side early exit of the seed-IK solver (reference _calculate_exit_condition, seed_ik_solver.py:452-468): sets
 * *stop_flag = 1 when at least `needed` of the num_problems problems have a converged seed in success [P, S].  Launches of
 * curobo_hip_seed_ik_iterate that are given the flag (optional, NULL = always run) return at once when it is set and count
 * themselves in *blocks_run otherwise, so the host can enqueue every block of iterations without a round trip. */
int curobo_hip_seed_ik_batch_status(const uint8_t *success, int num_problems, int num_seeds, int needed,
                                    int32_t *stop_flag, curobo_hip_stream_t stream);

/* Ranking of the seeds of every problem in one launch (ties -> lower seed index: the order of a stable sort).
 * curobo_hip_seed_ik_select: reference SeedIKSolver._select_top_solutions (seed_ik_solver.py:522-572): success =
 * position / orientation error under the tolerances (and strictly inside the limits), cost = position + orientation
 * error (+ start_cspace_dist_weight |q - current_position|) + 1e10 for failures; the return_seeds best, best first.
 * curobo_hip_ik_rank: reference IKSolver._get_result (solver_ik.py:440-580): feasible = no self collision, no
 * joint-limit cost, no scene collision (scene_distance [P, S, num_scene_columns], NULL = no scene), success = feasible
 * and EVERY tool frame within the thresholds (position_distance / rotation_distance / goalset_idx [P, S, T]), ranked by
 * cost + 1e16 for failures; out_position_error / out_rotation_error [P, k] = the largest error over the tool frames,
 * out_goalset_index [P, k, T] (ABI 6: one column per tool frame, as the reference returns it).  num_seeds <= 1024. */
int curobo_hip_seed_ik_select(
    uint8_t *out_success, float *out_solution, float *out_position_error, float *out_orientation_error,
    const float *joint_position, const float *position_error, const float *orientation_error, const float *limit_lower,
    const float *limit_upper, const float *current_position, float position_tolerance, float orientation_tolerance,
    float start_cspace_dist_weight, int check_limits, int num_problems, int num_seeds, int dof, int return_seeds,
    curobo_hip_stream_t stream);
int curobo_hip_ik_rank(
    uint8_t *out_success, float *out_solution, float *out_position_error, float *out_rotation_error, float *out_cost,
    int64_t *out_seed_index, int64_t *out_goalset_index, const float *joint_position, const float *cost,
    const float *position_distance, const float *rotation_distance, const float *self_collision_distance,
    const float *cspace_cost, const float *scene_distance, const int32_t *goalset_idx, float position_threshold,
    float rotation_threshold, int num_problems, int num_seeds, int dof, int num_tool_frames, int num_scene_columns,
    int return_seeds, int seed_offset, curobo_hip_stream_t stream);

/* Local stage of the seed-parallel arg-min exchange in one launch: out_rows[p] = (min over the seeds of cost[p, :], seed_offset
 * + the first index that attains it (as fp32), payload[p, index, :]), [num_problems, 2 + payload_width].  Reference
 * single-GPU equivalent: solver_ik.py:503-515, solver_trajopt.py:469-484. */
int curobo_hip_argmin_rows(float *out_rows, const float *cost, const float *payload, int num_problems, int num_seeds,
                           int payload_width, int seed_offset, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- optimization: MPPI update
 * reference: optim/particle/mppi.py:201-313 + jit helpers :615-757 (pure torch, DIAG_A
 * covariance).  costs [problems, particles, cost_horizon] (cost_horizon may be 1 for totals),
 * gamma_seq [cost_horizon], actions [problems, particles, action_horizon, action_dim],
 * mean [problems, action_horizon, action_dim], cov / new_cov / new_scale_tril
 * [problems, 1, action_dim].  w = softmax(-(sum_h gamma_h cost_h / gamma_0) / beta);
 * best_traj (optional) = the action sequence of arg-max w (first index); weights (optional)
 * [problems, particles]. */
int curobo_hip_mppi_update_distribution(
    float *new_mean, float *new_cov, float *new_scale_tril, float *best_traj, float *weights,
    const float *costs, const float *gamma_seq, const float *actions, const float *mean,
    const float *cov, int num_problems, int num_particles, int cost_horizon, int action_horizon,
    int action_dim, float beta, float step_size_mean, float step_size_cov, float kappa,
    curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- fused rollout
 * One launch for the data path of RobotRollout.evaluate_action + cost.backward
 * (reference rollout/rollout_robot.py:252-263,537-587, optim/components/gradient_opt_core.py
 * :445-480): B-spline knots -> joint positions -> FK -> collision spheres -> self collision +
 * (swept) scene collision -> out_cost[b] (sum over the padded horizon) and
 * out_grad_knots[b, n_knots, dof] = d out_cost / d knots.  It replaces the launch sequence
 * bspline forward, kinematics forward, self collision, sphere-obstacle collision, cost sum,
 * kinematics backward, bspline backward of the entry points above with identical arithmetic; all
 * intermediates stay in LDS.  out_position [b, h, dof] and out_robot_spheres [b, h, s, 4] are
 * optional (NULL = not materialised).  pair_locations / self_collision_weight NULL = no self
 * collision term; scene / scene_collision_weight NULL = no scene term.  Returns
 * CUROBO_HIP_ERR_INVALID when one trajectory's working set does not fit in 160 KB of LDS (use the
 * unfused entry points then). */
int curobo_hip_rollout_trajectory_fused(
    float *out_cost, float *out_grad_knots, float *out_position, float *out_robot_spheres,
    const float *u_position, const float *start_position, const float *start_velocity,
    const float *start_acceleration, const float *start_jerk, const float *goal_position,
    const float *goal_velocity, const float *goal_acceleration, const float *goal_jerk,
    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,
    const uint8_t *use_implicit_goal_state, const float *fixed_transform,
    const float *robot_spheres, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const float *joint_offset_map, const float *sphere_padding,
    const float *self_collision_weight, const int16_t *pair_locations,
    const curobo_hip_scene *scene, const float *scene_collision_weight,
    const float *activation_distance, const float *speed_dt, const int32_t *env_query_idx,
    int num_envs, int use_multi_env, int batch_size, int padded_horizon, int dof, int n_knots,
    int bspline_degree, int num_links, int num_spheres, int num_collision_pairs,
    int link_chain_len, int sweep_steps, int enable_speed_metric, int32_t *dispatch_ws,
    int dispatch_phase, const uint32_t *self_lane_lists, int self_lane_len, curobo_hip_stream_t stream);

/* The pair list of a robot dealt to lanes, for the self-collision pass of the fused trajectory kernels (optional:
 * self_lane_lists NULL = the pass walks pair_locations as the reference kernel does, self_collision_kernel.cuh:19-111).
 * Every pair is given to one of its two spheres so that the longest per-sphere list is as short as possible; a lane
 * then keeps its own sphere in registers and reads one partner per pair.  Same arg-max, same bits.  HOST pointers in,
 * HOST words out ((len0 + len1) * 64 of them; 64 * (num_collision_pairs / 64 + 2) words of capacity always suffice for
 * robots of up to 64 spheres); the return value (len0 | len1 << 16) is the self_lane_len argument; 0 = this robot is
 * outside the form (pass NULL); < 0 = bad arguments.  Copy the words to the device once per robot. */
int curobo_hip_self_lane_lists_host(uint32_t *out_lists_host, int capacity_words, const int16_t *pair_locations_host,
                                    int num_collision_pairs, int num_spheres);

/* Longest-first dispatch workspace of the fused trajectory kernels (optional; no reference
 * counterpart: the reference launches one thread per sphere, its work per thread block is
 * uniform).  dispatch_ws = device int32 [curobo_hip_rollout_dispatch_ws_size(batch)], set up once by
 * curobo_hip_rollout_dispatch_ws_init; the caller alternates dispatch_phase 0, 1, 0, ... between
 * consecutive launches on the same batch (each launch measures its workgroup durations and one
 * workgroup sorts the previous launch's into the order the next launch uses).  Outputs are
 * identical with and without it (NULL = blockIdx order); only the tail of the launch shortens. */
int curobo_hip_rollout_dispatch_ws_size(int batch_size);
int curobo_hip_rollout_dispatch_ws_init(int32_t *dispatch_ws, int batch_size, curobo_hip_stream_t stream);

/* Optional cost terms of the full trajopt task (reference content/configs/task/trajopt/
 * lbfgs_bspline_trajopt.yml: tool_pose_cfg + cspace_cfg with cost_type STATE) for
 * curobo_hip_rollout_trajopt_fused.  Pointer meaning = curobo_hip_tool_pose_distance and
 * curobo_hip_cspace_state_cost.  Set n_tool_frames = 0 / cspace_weight = NULL to switch a term off.
 * Optional metric outputs (NULL = skip): out_pose_distance [b, h, T, 2], out_position_distance,
 * out_rotation_distance, out_goalset_idx [b, h, T], out_cspace_cost [b, h, dof]. */
typedef struct curobo_hip_trajopt_terms {
  float *out_pose_distance, *out_position_distance, *out_rotation_distance;
  int32_t *out_goalset_idx;
  const float *goal_position, *goal_quat;
  const int32_t *idxs_goal;
  const float *position_orientation_weight;
  const float *terminal_pose_axes_weight_factor, *non_terminal_pose_axes_weight_factor;
  const float *terminal_pose_convergence_tolerance, *non_terminal_pose_convergence_tolerance;
  const uint8_t *project_distance_to_goal;
  const int16_t *tool_frame_map;
  int32_t n_tool_frames, num_goalset, rotation_method;
  float *out_cspace_cost;
  const float *state_dt, *target_joint_position;
  const int32_t *idxs_target_joint_position;
  const float *p_b, *v_b, *a_b, *j_b, *effort_b;
  const float *cspace_weight, *cspace_activation_distance, *squared_l2_regularization_weights;
  const float *cspace_target_weight, *cspace_non_terminal_weight_factor, *cspace_target_dof_weight;
  int32_t retime_weights, retime_regularization_weights;
  /* joint-torque limits: the effort terms of the c-space STATE cost on tau = RNEA(q, qd, qdd) (reference
   * cost/wp_cspace_state.py + kernels/dynamics/rnea_*_kernel.cuh), inverse dynamics and its VJP inside the launch.
   * Pointers as in curobo_hip_launch_rnea_forward; needs curobo_hip_rollout_trajopt_fused_torque_fits(...) != 0 */
  const float *link_masses_com, *link_inertias, *gravity;
  const int16_t *level_links;
  int32_t use_torque_limits;
} curobo_hip_trajopt_terms;

/* curobo_hip_rollout_trajectory_fused plus the optional terms above: the whole reference trajopt
 * rollout (B-spline -> FK -> tool pose + c-space STATE + self + swept scene collision -> cost and
 * gradient to the knots) in one launch; the velocity / acceleration / jerk samples and their
 * gradients live in LDS as well.  terms may be NULL (then identical to the function above). */
int curobo_hip_rollout_trajopt_fused(
    float *out_cost, float *out_grad_knots, float *out_position, float *out_robot_spheres,
    const float *u_position, const float *start_position, const float *start_velocity,
    const float *start_acceleration, const float *start_jerk, const float *goal_position,
    const float *goal_velocity, const float *goal_acceleration, const float *goal_jerk,
    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,
    const uint8_t *use_implicit_goal_state, const float *fixed_transform,
    const float *robot_spheres, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const float *joint_offset_map, const float *sphere_padding,
    const float *self_collision_weight, const int16_t *pair_locations,
    const curobo_hip_scene *scene, const float *scene_collision_weight,
    const float *activation_distance, const float *speed_dt, const int32_t *env_query_idx,
    int num_envs, int use_multi_env, int batch_size, int padded_horizon, int dof, int n_knots,
    int bspline_degree, int num_links, int num_spheres, int num_collision_pairs,
    int link_chain_len, int sweep_steps, int enable_speed_metric, int32_t *dispatch_ws,
    int dispatch_phase, const uint32_t *self_lane_lists, int self_lane_len, const curobo_hip_trajopt_terms *terms,
    curobo_hip_stream_t stream);

/* 1 if the torque-limit terms fit (they borrow LDS regions that are dead when the inverse dynamics runs) */
int curobo_hip_rollout_trajopt_fused_torque_fits(
    int padded_horizon, int dof, int num_links, int num_spheres, int num_collision_pairs,
    int link_chain_len, int num_obstacles);

int curobo_hip_rollout_trajopt_fused_lds_bytes(
    int padded_horizon, int dof, int num_links, int num_spheres, int num_collision_pairs,
    int link_chain_len, int num_obstacles, int with_cspace_terms);

/* LDS bytes one trajectory needs in curobo_hip_rollout_trajectory_fused (host-side query, no GPU
 * work); the fused entry point is usable when this is <= 163840.  num_obstacles = max_cuboids +
 * max_voxel_grids of the scene (0 without a scene term). */
int curobo_hip_rollout_trajectory_fused_lds_bytes(
    int padded_horizon, int dof, int num_links, int num_spheres, int num_collision_pairs,
    int link_chain_len, int num_obstacles);

/* Horizon-1 (IK / teleport) rollout in one launch: q[b, dof] -> FK -> tool-pose goal-set cost
 * (curobo_hip_tool_pose_distance semantics, terminal weights) + joint-limit term of the c-space
 * cost (weight[0], activation_distance[0], limits p_b[2, dof]) + self collision + scene collision
 * (no sweep) -> out_cost[b] and out_grad_q[b, dof].  Replaces the launch sequence kinematics
 * forward, tool_pose_distance, cspace_position_cost, self_collision_distance,
 * sphere_obstacle_collision, kinematics backward, rollout_point_aggregate (reference: RobotRollout
 * with StateFromPositionTeleport and content/configs/task/ik/lbfgs_ik.yml).  Optional outputs
 * (NULL = skip): out_pose_distance [b, T, 2], out_position_distance / out_rotation_distance /
 * out_goalset_idx [b, T], out_link_pos [b, T, 3], out_link_quat [b, T, 4] (wxyz),
 * out_robot_spheres [b, S, 4], out_cspace_cost [b, dof].  env_query_idx [b] (use_multi_env: scene environment; num_envs > 1:
 * sphere set) must be constant over aligned runs of 16 configurations -- the seeds of one problem -- which share a
 * workgroup and its staged tables; NULL / 1 / 0 = one environment. */
int curobo_hip_rollout_ik_fused(
    float *out_cost, float *out_grad_q, float *out_pose_distance, float *out_position_distance,
    float *out_rotation_distance, int32_t *out_goalset_idx, float *out_link_pos,
    float *out_link_quat, float *out_robot_spheres, float *out_cspace_cost, const float *q,
    const float *goal_position, const float *goal_quat, const int32_t *idxs_goal,
    const float *position_orientation_weight, const float *terminal_pose_axes_weight_factor,
    const float *terminal_pose_convergence_tolerance, const uint8_t *project_distance_to_goal,
    int num_goalset, int rotation_method, const float *p_b, const float *cspace_weight,
    const float *cspace_activation_distance, const float *fixed_transform,
    const float *robot_spheres, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_sphere_map,
    const int16_t *link_chain_data, const int16_t *link_chain_offsets,
    const float *joint_offset_map, const float *sphere_padding,
    const float *self_collision_weight, const int16_t *pair_locations,
    const curobo_hip_scene *scene, const float *scene_collision_weight,
    const float *activation_distance, int batch_size, int dof, int num_links, int n_tool_frames,
    int num_spheres, int num_collision_pairs, int link_chain_len, const int32_t *env_query_idx, int num_envs,
    int use_multi_env, curobo_hip_stream_t stream);

/* LDS bytes of one 16-configuration workgroup of curobo_hip_rollout_ik_fused (usable when <= 163840) */
int curobo_hip_rollout_ik_fused_lds_bytes(int dof, int num_links, int num_spheres,
                                          int num_collision_pairs, int link_chain_len,
                                          int num_obstacles);

/* Development hook: device buffer [batch, 16] (int64) that receives 100 MHz wall-clock stamps at
 * the phase boundaries (start, tables+spline, FK, costs+VJP, end) of every fused launch; NULL
 * (default) turns it off.  Used by tools/profile_fused.py. */
int curobo_hip_rollout_fused_set_profile_buffer(int64_t *device_buffer);
/* Same hook for a sequence of launches (e.g. those recorded into a hipGraph): launch k after this
 * call stamps into block k = device_buffer + k * block_rows * 16; launches beyond n_blocks (or with
 * more than block_rows trajectories) are not stamped.  NULL ends the sequence.  bench.py uses it to
 * time the rollout launches inside the replayed graph (stamp 0 = workgroup start, 4 = end). */
int curobo_hip_rollout_fused_set_profile_sequence(int64_t *device_buffer, int n_blocks, int block_rows);

/* Compile-time shapes of the fused trajectory launches (csrc/fused_shapes.hpp): for the robots / horizons listed there the
 * library holds instantiations whose every dimension is a compile-time constant (the reference compiles its kernels per
 * robot with NVRTC templates: kinematics_forward_kernel.cuh:126 N_LINKS, cuda_core_backend/kernel_cache.py:161-235); a launch
 * takes one only when ALL its dimensions match, otherwise the generic kernel -- same results, bit for bit.
 * curobo_hip_rollout_fused_shape_id: host-side query (no GPU work), the id (>= 1) of the shape a launch with these arguments
 * runs, 0 = the generic kernel.  self_lane_len = the value curobo_hip_self_lane_lists_host returned (0 = no lane lists);
 * num_collision_pairs = 0 when the self-collision term is off; kinds = 1 cuboids | 2 voxel grids (7 = analytic primitives).
 * plain_launch = 1: the launch form of an optimiser iteration -- self + scene collision with the speed metric, one environment,
 * a dispatch workspace, no materialised outputs, no profile stamps -- which some shapes also hold as compile-time facts.
 * curobo_hip_rollout_fused_set_shapes_enabled(0) makes every launch take the generic kernel (tests, A/B timing). */
int curobo_hip_rollout_fused_shape_id(int padded_horizon, int n_knots, int dof, int num_links, int num_spheres,
                                      int num_collision_pairs, int link_chain_len, int self_lane_len, int max_cuboids,
                                      int max_voxel_grids, int bspline_degree, int sweep_steps, int kinds,
                                      int with_trajopt_terms, int plain_launch);
int curobo_hip_rollout_fused_set_shapes_enabled(int enabled);
/* Shapes compiled at run time (curobo_amd/backends/fused_jit.py: hipcc on csrc/rollout_fused.hip with the shape on the command
 * line, as the reference compiles its kernels per robot with NVRTC): `launcher` = the object's curobo_fused_jit_launch,
 * args_bytes = its curobo_fused_jit_args_bytes() (an object built from other sources is refused).  Registered shapes are tried
 * before the built-in table and report ids >= 100 from curobo_hip_rollout_fused_shape_id.
 * curobo_hip_rollout_fused_threads returns the workgroup size a launch with these dimensions uses (a shape holds it as a
 * constant); num_obstacles = max_cuboids + max_voxel_grids.  Host-side, no GPU work. */
int curobo_hip_rollout_fused_register_shape(void *launcher, int args_bytes);
int curobo_hip_rollout_fused_threads(int padded_horizon, int dof, int num_links, int num_spheres, int num_collision_pairs,
                                     int link_chain_len, int self_lane_len, int num_obstacles, int with_trajopt_terms);

/* ---------------------------------------------------------------- trajectory: B-spline
 * reference: cuda_core_backend/trajectory.py:28-204, pybind/trajectory_bindings.cpp:133-142
 * kernels:   kernels/trajectory/bspline/bspline_kernel.cuh:81-151,332-380
 * `horizon` of the forward launch is the padded horizon (out_position.shape[1]).
 */
int curobo_hip_launch_bspline_interpolation_forward_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk,
    float *out_dt, const float *u_position, const float *start_position,
    const float *start_velocity, const float *start_acceleration, const float *start_jerk,
    const float *goal_position, const float *goal_velocity, const float *goal_acceleration,
    const float *goal_jerk, const int32_t *start_idx, const int32_t *goal_idx,
    const float *traj_dt, const uint8_t *use_implicit_goal_state, int batch_size, int horizon,
    int dof, int n_knots, int bspline_degree, curobo_hip_stream_t stream);

int curobo_hip_launch_bspline_interpolation_backward_kernel(
    float *out_grad_position, const float *grad_position, const float *grad_velocity,
    const float *grad_acceleration, const float *grad_jerk, const float *traj_dt,
    const int32_t *dt_idx, const uint8_t *use_implicit_goal_state, int batch_size,
    int padded_horizon, int dof, int n_knots, int bspline_degree, int use_direct_polynomial,
    curobo_hip_stream_t stream);

/* reference: cuda_core_backend/trajectory.py:207-306, kernel bspline_kernel.cuh:221-270.
 * One interpolation_dt[1] for all trajectories, interpolation_horizon[b] per trajectory; outputs
 * are [batch, max_out_tsteps, dof]; points past a trajectory's horizon repeat its last sample.
 * knot_dt is accepted and ignored, as in the reference kernel. */
int curobo_hip_launch_bspline_interpolation_single_dt_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk,
    float *out_dt, const float *knots, const float *knot_dt, const float *start_position,
    const float *start_velocity, const float *start_acceleration, const float *start_jerk,
    const float *goal_position, const float *goal_velocity, const float *goal_acceleration,
    const float *goal_jerk, const int32_t *start_idx, const int32_t *goal_idx,
    const float *interpolation_dt, const uint8_t *use_implicit_goal_state,
    const int32_t *interpolation_horizon, int batch_size, int max_out_tsteps, int dof,
    int n_knots, int bspline_degree, curobo_hip_stream_t stream);

/* Legacy control spaces (reference cuda_core_backend/trajectory.py:309-556; kernels
 * kernels/trajectory/legacy/differentiation_position_kernel.cuh:15-370 with use_stencil = true,
 * legacy/integration_acceleration_kernel.cuh:8-135).
 * POSITION: u_position [batch, horizon-4, dof] -> position/velocity/acceleration/jerk
 * [batch, horizon, dof] by five-point stencils over the start-extrapolated, goal-replicated
 * position sequence; horizon >= 9.  goal_velocity / goal_acceleration are accepted and unused.
 * ACCELERATION: u_acc [batch, horizon, dof], traj_dt [horizon] (indexed by step); use_rk2 selects
 * nothing (both reference kernels run the same recursion). */
int curobo_hip_launch_differentiation_position_forward_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk,
    float *out_dt, const float *u_position, const float *start_position,
    const float *start_velocity, const float *start_acceleration, const float *goal_position,
    const float *goal_velocity, const float *goal_acceleration, const int32_t *start_idx,
    const int32_t *goal_idx, const float *traj_dt, const uint8_t *use_implicit_goal_state,
    int batch_size, int horizon, int dof, curobo_hip_stream_t stream);

int curobo_hip_launch_differentiation_position_backward_kernel(
    float *out_grad_position, const float *grad_position, const float *grad_velocity,
    const float *grad_acceleration, const float *grad_jerk, const float *traj_dt,
    const int32_t *dt_idx, const uint8_t *use_implicit_goal_state, int batch_size, int horizon,
    int dof, curobo_hip_stream_t stream);

int curobo_hip_launch_integration_acceleration_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk,
    const float *u_acc, const float *start_position, const float *start_velocity,
    const float *start_acceleration, const int32_t *start_idx, const float *traj_dt,
    int batch_size, int horizon, int dof, int use_rk2, curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- optimization
 * reference: cuda_core_backend/optimization.py:27-260, pybind/optimization_bindings.cpp:14-75
 * kernels:   kernels/optimization/lbfgs/lbfgs_step_kernel.cuh:18-199,
 *            kernels/optimization/line_search/line_search_kernel.cuh:27-155
 */
int curobo_hip_launch_lbfgs_step(
    float *step_vec, float *rho_buffer, float *y_buffer, float *s_buffer, const float *q,
    const float *grad_q, float *x_0, float *grad_0, float epsilon, int batch_size, int history_m,
    int v_dim, int stable_mode, int use_shared_buffers, curobo_hip_stream_t stream);

int curobo_hip_launch_line_search(
    float *best_cost, float *best_action, int16_t *best_iteration, int16_t *current_iteration,
    uint8_t *converged_global, int convergence_iteration, float cost_delta_threshold,
    float cost_relative_threshold, float *exploration_cost, float *exploration_action,
    float *exploration_gradient, int32_t *exploration_idx, float *selected_cost,
    float *selected_action, float *selected_gradient, int32_t *selected_idx,
    const float *search_cost, const float *search_action, const float *search_gradient,
    const float *step_direction, const float *search_magnitudes, float armijo_threshold_c_1,
    float curvature_threshold_c_2, int strong_wolfe, int approx_wolfe, int n_linesearch,
    int opt_dim, int batchsize, curobo_hip_stream_t stream);

/* Line-search candidates, fused (extension; reference: torch ops in
 * optim/gradient/line_search_strategy.py:134-204 `_prepare_search_points`, :281-325):
 *   scale = max(1, max_v |d_v| / action_step_max[v % action_dim])  (if apply_step_scale)
 *   step_direction_out = d / scale;  x_set[b,k,:] = x[b,:] + search_magnitudes[k] * d / scale */
int curobo_hip_prepare_search_points(
    float *x_set, float *step_direction_out, const float *x, const float *step_direction,
    const float *action_step_max, const float *search_magnitudes, int batchsize, int n_linesearch,
    int opt_dim, int action_dim, int apply_step_scale, curobo_hip_stream_t stream);

/* The optimiser side of one L-BFGS iteration in ONE launch (opt_dim <= 128): launch_line_search,
 * then launch_lbfgs_step from the chosen exploration point, then prepare_search_points for the
 * next iteration -- the same arithmetic as the three entry points above run back to back
 * (reference optim/gradient/lbfgs.py:156-265, gradient_opt_core.py:255-480), with the exploration
 * point and the new direction handed over in registers.  search_action (the candidate set x_set)
 * and step_direction_scaled are read by the line search and then overwritten with the next
 * iteration's candidates / scaled direction.  step_vec receives the unscaled direction.
 * overlapped != 0: the launch shares the GPU with other work (the seed shards of optim/pipelined.py run their rollouts
 * next to it): wavefront-sized problems then take ONE wavefront and no LDS each instead of a 256-lane workgroup with
 * the history staged through LDS -- 1 us more on an idle GPU, 5 us less per iteration between busy rollout
 * workgroups, whose LDS and registers the workgroup form has to wait for.  Same results bit for bit. */
int curobo_hip_launch_lbfgs_iteration_tail(
    float *best_cost, float *best_action, int16_t *best_iteration, int16_t *current_iteration,
    uint8_t *converged_global, int convergence_iteration, float cost_delta_threshold,
    float cost_relative_threshold, float *exploration_cost, float *exploration_action,
    float *exploration_gradient, int32_t *exploration_idx, float *selected_cost,
    float *selected_action, float *selected_gradient, int32_t *selected_idx,
    const float *search_cost, float *search_action, const float *search_gradient,
    float *step_direction_scaled, const float *search_magnitudes, float armijo_threshold_c_1,
    float curvature_threshold_c_2, int strong_wolfe, int approx_wolfe, int n_linesearch,
    int opt_dim, int batchsize, float *step_vec, float *rho_buffer, float *y_buffer,
    float *s_buffer, float *x_0, float *grad_0, float epsilon, int history_m, int stable_mode,
    const float *action_step_max, int action_dim, int apply_step_scale, int overlapped,
    curobo_hip_stream_t stream);

/* ---------------------------------------------------------------- rollout glue
 * Per-trajectory cost sum (reference rollout/metrics.py:233-265 + util/tensor_util.py:104:
 * torch cat + sum): out[b] = sum_h( self_cost[b,h] + sum_s scene_cost[b,h,s] ), one wavefront
 * per trajectory with a wave64 shuffle reduction.  Either input may be NULL. */
int curobo_hip_trajectory_cost_sum(float *out_cost, const float *self_cost,
                                   const float *scene_cost, int batch_size, int horizon,
                                   int num_spheres, curobo_hip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CUROBO_HIP_H */
