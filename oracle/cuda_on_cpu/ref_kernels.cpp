// ref_kernels.cpp -- C entry points that LAUNCH THE REFERENCE'S OWN CUDA KERNELS on the CPU (oracle/cuda_on_cpu,
// TEST INFRASTRUCTURE).  The kernel sources are included from /root/reference/curobo/_src/curobolib/kernels at build time
// (the three files with `extern __shared__` arrays through a one-line sed rewrite into oracle/_ref/gen, see the Makefile);
// launch geometry as in curobolib/backends/cuda_core_backend/kinematics_config.py:53-92 (geometry does not change results).
#include "simt.hpp"

#include "kinematics/kinematics_forward_kernel.cuh"

using namespace curobo::kinematics;

extern "C" int ref_kinematics_forward(float *link_pos, float *link_quat, float *batch_com, float *global_cumul, const float *q,
                                      const float *fixed_transform, const float *link_masses_com, const int8_t *joint_map_type,
                                      const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map,
                                      const float *joint_offset, int batch_size, int horizon, int num_links, int n_joints,
                                      int n_tool_frames, int compute_com) {
  const int threads_per_batch = 32, max_threads = 256;
  int bpb = std::min(MAX_FW_BATCH_PER_BLOCK, (48 * 1024) / (num_links * 12 * 4 * 2));
  if (bpb * threads_per_batch > max_threads) bpb = max_threads / threads_per_batch;
  bpb = std::max(1, std::min(bpb, batch_size));
  const dim3 grid((batch_size + bpb - 1) / bpb), block(bpb * threads_per_batch);
  const size_t smem = (size_t)bpb * num_links * 12 * 4 * 2;
  auto body = [&] {
    if (compute_com)
      kinematics_forward_kernel<-1, true>(link_pos, link_quat, batch_com, global_cumul, q, fixed_transform, link_masses_com,
                                          joint_map_type, joint_map, link_map, tool_frame_map, joint_offset, batch_size, horizon,
                                          num_links, n_joints, n_tool_frames);
    else
      kinematics_forward_kernel<-1, false>(link_pos, link_quat, batch_com, global_cumul, q, fixed_transform, link_masses_com,
                                           joint_map_type, joint_map, link_map, tool_frame_map, joint_offset, batch_size, horizon,
                                           num_links, n_joints, n_tool_frames);
  };
  cuoc::launch(grid, block, smem, body);
  return 0;
}

// geometry of calculate_forward_config for `threads_per_batch` output threads per configuration
static void forward_geometry(int batch_size, int num_links, int threads_per_batch, dim3 *grid, dim3 *block, size_t *smem) {
  const int max_threads = 256;
  int bpb = std::min(MAX_FW_BATCH_PER_BLOCK, (48 * 1024) / (num_links * 12 * 4 * 2));
  if (bpb * threads_per_batch > max_threads) bpb = max_threads / threads_per_batch;
  bpb = std::max(1, std::min(bpb, batch_size));
  *grid = dim3((batch_size + bpb - 1) / bpb);
  *block = dim3(bpb * threads_per_batch);
  *smem = (size_t)bpb * num_links * 12 * 4 * 2;
}

// kinematics_forward_spheres_kernel<-1, 32, true, COM> and kinematics_forward_spheres_jacobian_kernel<-1, 32, true, COM>
// (jacobian != NULL), reference launch: cuda_core_backend/kinematics.py:90-290
extern "C" int ref_kinematics_forward_spheres(
    float *link_pos, float *link_quat, float *b_robot_spheres, float *batch_com, float *jacobian, float *global_cumul, const float *q,
    const float *fixed_transform, const float *robot_spheres, const float *link_masses_com, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_sphere_map,
    const int16_t *link_chain_data, const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const bool *joint_affects_endeffector, const float *joint_offset,
    const int32_t *env_query_idx, int batch_size, int horizon, int nspheres, int num_envs, int num_links, int n_joints,
    int n_tool_frames, int compute_com) {
  dim3 grid, block;
  size_t smem;
  forward_geometry(batch_size, num_links, 32, &grid, &block, &smem);
  auto body = [&] {
    if (jacobian) {
      if (compute_com)
        kinematics_forward_spheres_jacobian_kernel<-1, 32, true, true>(
            link_pos, link_quat, b_robot_spheres, batch_com, jacobian, global_cumul, q, fixed_transform, robot_spheres, link_masses_com,
            joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map, link_chain_data, link_chain_offsets, joint_links_data,
            joint_links_offsets, joint_affects_endeffector, joint_offset, env_query_idx, batch_size, horizon, nspheres, num_envs,
            num_links, n_joints, n_tool_frames);
      else
        kinematics_forward_spheres_jacobian_kernel<-1, 32, true, false>(
            link_pos, link_quat, b_robot_spheres, batch_com, jacobian, global_cumul, q, fixed_transform, robot_spheres, link_masses_com,
            joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map, link_chain_data, link_chain_offsets, joint_links_data,
            joint_links_offsets, joint_affects_endeffector, joint_offset, env_query_idx, batch_size, horizon, nspheres, num_envs,
            num_links, n_joints, n_tool_frames);
    } else {
      if (compute_com)
        kinematics_forward_spheres_kernel<-1, 32, true, true>(
            link_pos, link_quat, b_robot_spheres, batch_com, global_cumul, q, fixed_transform, robot_spheres, link_masses_com,
            joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map, joint_offset, env_query_idx, batch_size, horizon,
            nspheres, num_envs, num_links, n_joints, n_tool_frames);
      else
        kinematics_forward_spheres_kernel<-1, 32, true, false>(
            link_pos, link_quat, b_robot_spheres, batch_com, global_cumul, q, fixed_transform, robot_spheres, link_masses_com,
            joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map, joint_offset, env_query_idx, batch_size, horizon,
            nspheres, num_envs, num_links, n_joints, n_tool_frames);
    }
  };
  cuoc::launch(grid, block, smem, body);
  return 0;
}
