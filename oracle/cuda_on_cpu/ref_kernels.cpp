// ref_kernels.cpp -- C entry points that LAUNCH THE REFERENCE'S OWN CUDA KERNELS on the CPU (oracle/cuda_on_cpu,
// TEST INFRASTRUCTURE).  The kernel sources are included from /root/reference/curobo/_src/curobolib/kernels at build time
// (the files with `extern __shared__` arrays and the warp reduction through two one-line sed rewrites into a temporary
// directory, see the Makefile);
// launch geometry as in curobolib/backends/cuda_core_backend/kinematics_config.py:53-92 (geometry does not change results).
#include "simt.hpp"

#include <type_traits>

#include "kinematics/kinematics_forward_kernel.cuh"
#include "geometry/self_collision/self_collision_kernel.cuh"
#include "kinematics/kinematics_backward_kernel.cuh"
#include "trajectory/bspline/bspline_kernel.cuh"
#include "optimization/line_search/line_search_kernel.cuh"
#include "trajectory/legacy/differentiation_position_kernel.cuh"
#include "trajectory/legacy/integration_acceleration_kernel.cuh"
#include "dynamics/rnea_forward_kernel.cuh"
#include "dynamics/rnea_backward_kernel.cuh"
#include "optimization/lbfgs/lbfgs_step_kernel.cuh"

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

// self_collision_max_distance_kernel<false> (one block per point), reference launch: cuda_core_backend/geometry.py:98-144
extern "C" int ref_self_collision_distance(float *out_distance, float *out_vec, float *pair_distance, uint8_t *sparse_index,
                                           const float *robot_spheres, const float *sphere_padding, const float *weight,
                                           int16_t *pair_locations, int batch_size, int horizon, int nspheres, int num_collision_pairs,
                                           int max_threads_per_block, int compute_grad) {
  int threads = std::min(max_threads_per_block, num_collision_pairs);
  threads = std::min(((threads + 31) / 32) * 32, max_threads_per_block);
  const dim3 grid(batch_size * horizon), block(threads);
  auto body = [&] {
    curobo::geometry::self_collision::self_collision_max_distance_kernel<false>(
        out_distance, out_vec, pair_distance, sparse_index, robot_spheres, sphere_padding, weight, pair_locations, batch_size, horizon,
        nspheres, num_collision_pairs, compute_grad != 0);
  };
  cuoc::launch(grid, block, (size_t)16 * nspheres, body);
  return 0;
}

// kinematics_backward_kernel<float, float, MAX_JOINTS, true (warp reduce), COM, false>, reference launch:
// cuda_core_backend/kinematics.py:293-379, geometry of kinematics_config.py:94-175 (num_spheres < 5000 branch)
template <int16_t MAXJ>
static void backward_launch(dim3 grid, dim3 block, size_t smem, bool com, float *grad_q, const float *g_pos, const float *g_quat,
                            const float *g_sph, const float *g_com, const float *b_com, const float *g_jac, const float *cumul,
                            const float *robot_spheres, const float *masses, const int8_t *jtype, const int16_t *jmap,
                            const int16_t *lmap, const int16_t *tmap, const int16_t *smap, const int32_t *env, const int16_t *lcd,
                            const int16_t *lco, const int16_t *jld, const int16_t *jlo, const bool *jae, const float *joff, int B, int H,
                            int S, int L, int J, int T, int E, int tpb) {
  auto body = [&] {
    if (com)
      kinematics_backward_kernel<float, float, MAXJ, true, true, false>(grad_q, g_pos, g_quat, g_sph, g_com, b_com, g_jac, cumul,
          robot_spheres, masses, jtype, jmap, lmap, tmap, smap, env, lcd, lco, jld, jlo, jae, joff, B, H, S, L, J, T, E, tpb);
    else
      kinematics_backward_kernel<float, float, MAXJ, true, false, false>(grad_q, g_pos, g_quat, g_sph, g_com, b_com, g_jac, cumul,
          robot_spheres, masses, jtype, jmap, lmap, tmap, smap, env, lcd, lco, jld, jlo, jae, joff, B, H, S, L, J, T, E, tpb);
  };
  cuoc::launch(grid, block, smem, body);
}

// the same launch with the Jacobian-gradient branch compiled in (COMPUTE_JACOBIAN_GRAD = true): d/dq of <grad_jacobian, J(q)>
// is added to the pose / sphere terms
extern "C" int ref_kinematics_backward_jacobian(
    float *grad_q, const float *grad_link_pos, const float *grad_link_quat, const float *grad_spheres, const float *grad_jacobian,
    const float *global_cumul, const float *robot_spheres, const float *link_masses_com, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_sphere_map,
    const int32_t *env_query_idx, const int16_t *link_chain_data, const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const bool *joint_affects_endeffector, const float *joint_offset, int batch_size, int horizon,
    int nspheres, int num_links, int n_joints, int n_tool_frames, int num_envs) {
  const int max_threads = 128, tpb = 32;
  int bpb = std::min(MAX_BW_BATCH_PER_BLOCK, (48 * 1024) / (num_links * 12 * 4));
  if (bpb * tpb > max_threads) bpb = max_threads / tpb;
  bpb = std::max(1, std::min(bpb, batch_size));
  const dim3 block(bpb * tpb), grid((batch_size * tpb + bpb * tpb - 1) / (bpb * tpb));
  const size_t smem = (size_t)bpb * num_links * 12 * 4;
  auto run = [&](auto maxj) {
    constexpr int16_t MAXJ = decltype(maxj)::value;
    cuoc::launch(grid, block, smem, [&] {
      kinematics_backward_kernel<float, float, MAXJ, true, false, true>(
          grad_q, grad_link_pos, grad_link_quat, grad_spheres, nullptr, nullptr, grad_jacobian, global_cumul, robot_spheres,
          link_masses_com, joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map, env_query_idx, link_chain_data,
          link_chain_offsets, joint_links_data, joint_links_offsets, joint_affects_endeffector, joint_offset, batch_size, horizon,
          nspheres, num_links, n_joints, n_tool_frames, num_envs, tpb);
    });
  };
  if (n_joints < 16) run(std::integral_constant<int16_t, 16>{});
  else if (n_joints < 64) run(std::integral_constant<int16_t, 64>{});
  else run(std::integral_constant<int16_t, 128>{});
  return 0;
}

extern "C" int ref_kinematics_backward(
    float *grad_q, const float *grad_link_pos, const float *grad_link_quat, const float *grad_spheres, const float *grad_com,
    const float *batch_com, const float *global_cumul, const float *robot_spheres, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const int32_t *env_query_idx, const int16_t *link_chain_data, const int16_t *link_chain_offsets,
    const int16_t *joint_links_data, const int16_t *joint_links_offsets, const bool *joint_affects_endeffector, const float *joint_offset,
    int batch_size, int horizon, int nspheres, int num_links, int n_joints, int n_tool_frames, int num_envs, int compute_com) {
  const int max_threads = 128, tpb = 32;
  int bpb = std::min(MAX_BW_BATCH_PER_BLOCK, (48 * 1024) / (num_links * 12 * 4));
  if (bpb * tpb > max_threads) bpb = max_threads / tpb;
  bpb = std::max(1, std::min(bpb, batch_size));
  const dim3 block(bpb * tpb), grid((batch_size * tpb + bpb * tpb - 1) / (bpb * tpb));
  const size_t smem = (size_t)bpb * num_links * 12 * 4;
#define CUOC_BW(MAXJ) backward_launch<MAXJ>(grid, block, smem, compute_com != 0, grad_q, grad_link_pos, grad_link_quat, grad_spheres, grad_com, \
    batch_com, nullptr, global_cumul, robot_spheres, link_masses_com, joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map,   \
    env_query_idx, link_chain_data, link_chain_offsets, joint_links_data, joint_links_offsets, joint_affects_endeffector, joint_offset,       \
    batch_size, horizon, nspheres, num_links, n_joints, n_tool_frames, num_envs, tpb)
  if (n_joints < 16) CUOC_BW(16);
  else if (n_joints < 64) CUOC_BW(64);
  else CUOC_BW(128);
#undef CUOC_BW
  return 0;
}

// the two-kernel form for long pair lists (Unitree G1: 162 k pairs): self_collision_max_block_kernel<false> over
// batch x horizon x num_blocks_per_batch blocks, then self_collision_max_reduce_kernel; cuda_core_backend/geometry.py:146-226.
// block_batch_max_value [n * num_blocks_per_batch] floats and block_batch_max_index [2 * n * num_blocks_per_batch] int16 are scratch.
extern "C" int ref_self_collision_distance_blocks(float *out_distance, float *out_vec, float *pair_distance, uint8_t *sparse_index,
                                                  const float *robot_spheres, const float *sphere_padding, const float *weight,
                                                  int16_t *pair_locations, float *block_batch_max_value, int16_t *block_batch_max_index,
                                                  int num_blocks_per_batch, int batch_size, int horizon, int nspheres,
                                                  int num_collision_pairs, int max_threads_per_block, int compute_grad) {
  namespace sc = curobo::geometry::self_collision;
  cuoc::launch(dim3(batch_size * horizon * num_blocks_per_batch), dim3(max_threads_per_block), (size_t)16 * nspheres, [&] {
    sc::self_collision_max_block_kernel<false>(out_vec, pair_distance, sparse_index, robot_spheres, sphere_padding, pair_locations,
                                               block_batch_max_value, block_batch_max_index, num_blocks_per_batch, batch_size, horizon,
                                               nspheres, num_collision_pairs);
  });
  cuoc::launch(dim3(batch_size * horizon), dim3(std::min(512, num_blocks_per_batch)), 0, [&] {
    sc::self_collision_max_reduce_kernel(out_distance, out_vec, pair_distance, sparse_index, robot_spheres, sphere_padding, weight,
                                         pair_locations, block_batch_max_value, block_batch_max_index, num_blocks_per_batch, batch_size,
                                         horizon, nspheres, num_collision_pairs, compute_grad != 0);
  });
  return 0;
}

// interpolate_bspline_kernel<float, Degree, MATRIX> and bspline_backward_kernel<Degree, float, MATRIX>; reference launch:
// cuda_core_backend/trajectory.py:21-207, geometry of trajectory_config.py:124-174
namespace bs = curobo::trajectory::bspline;
template <int DEG>
static void bspline_fwd(float *p, float *v, float *a, float *j, float *out_dt, const float *u, const float *sp, const float *sv,
                        const float *sa, const float *sj, const float *gp, const float *gv, const float *ga, const float *gj,
                        const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt, const uint8_t *implicit, int B, int PH,
                        int D, int K) {
  const int k_size = B * D * PH, threads = std::min(k_size, 128);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    bs::interpolate_bspline_kernel<float, DEG, bs::BasisBackend::MATRIX>(p, v, a, j, out_dt, u, sp, sv, sa, sj, gp, gv, ga, gj, start_idx,
                                                                        goal_idx, traj_dt, implicit, B, PH, D, K);
  });
}
template <int DEG>
static void bspline_bwd(float *out, const float *gp, const float *gv, const float *ga, const float *gj, const float *traj_dt,
                        const int32_t *dt_idx, const uint8_t *implicit, int B, int H, int D, int K) {
  const bs::BSplineBackwardLayout layout = bs::compute_bspline_backward_layout<DEG>(H, D, K);
  const int k_size = B * D * layout.threads_for_n_knots, threads = std::min(k_size, 128);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    bs::bspline_backward_kernel<DEG, float, bs::BasisBackend::MATRIX>(out, gp, gv, ga, gj, traj_dt, dt_idx, implicit, B, H, D, K);
  });
}
extern "C" int ref_bspline_forward(float *p, float *v, float *a, float *j, float *out_dt, const float *u, const float *sp, const float *sv,
                                   const float *sa, const float *sj, const float *gp, const float *gv, const float *ga, const float *gj,
                                   const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt, const uint8_t *implicit, int B,
                                   int PH, int D, int K, int degree) {
  if (degree == 3) bspline_fwd<3>(p, v, a, j, out_dt, u, sp, sv, sa, sj, gp, gv, ga, gj, start_idx, goal_idx, traj_dt, implicit, B, PH, D, K);
  else if (degree == 4) bspline_fwd<4>(p, v, a, j, out_dt, u, sp, sv, sa, sj, gp, gv, ga, gj, start_idx, goal_idx, traj_dt, implicit, B, PH, D, K);
  else if (degree == 5) bspline_fwd<5>(p, v, a, j, out_dt, u, sp, sv, sa, sj, gp, gv, ga, gj, start_idx, goal_idx, traj_dt, implicit, B, PH, D, K);
  else return 1;
  return 0;
}
extern "C" int ref_bspline_backward(float *out, const float *gp, const float *gv, const float *ga, const float *gj, const float *traj_dt,
                                    const int32_t *dt_idx, const uint8_t *implicit, int B, int H, int D, int K, int degree) {
  if (degree == 3) bspline_bwd<3>(out, gp, gv, ga, gj, traj_dt, dt_idx, implicit, B, H, D, K);
  else if (degree == 4) bspline_bwd<4>(out, gp, gv, ga, gj, traj_dt, dt_idx, implicit, B, H, D, K);
  else if (degree == 5) bspline_bwd<5>(out, gp, gv, ga, gj, traj_dt, dt_idx, implicit, B, H, D, K);
  else return 1;
  return 0;
}

// kernel_lbfgs_step<float, false, -1> and kernel_lbfgs_step_shared_memory<float, false, -1> (one block per problem, v_dim threads;
// reference launch: cuda_core_backend/optimization.py:138-260, LBFGSLaunchCfg: dynamic shared memory = history * 4 bytes for the
// alpha buffer, or (2 history v_dim + 2 history + 33) floats for the shared-memory form).  The first kernel hands its rho history
// from warp 0 to the whole block without a barrier; the build recipe spells that rendezvous out (Makefile).
extern "C" int ref_lbfgs_step(float *step_vec, float *rho_buffer, float *y_buffer, float *s_buffer, float *q, float *x_0, float *grad_0,
                              const float *grad_q, float epsilon, int batchsize, int m, int v_dim, int stable_mode, int shared_buffers) {
  if (m < 1 || m > 31 || v_dim < 1 || v_dim > 1024) return 1;
  if (shared_buffers) {
    const size_t smem = (size_t)(2 * m * v_dim + 2 * m + 32 + 1) * sizeof(float);
    cuoc::launch(dim3(batchsize), dim3(v_dim), smem, [&] {
      curobo::optimization::kernel_lbfgs_step_shared_memory<float, false, -1>(step_vec, rho_buffer, y_buffer, s_buffer, q, x_0, grad_0, grad_q,
                                                                              epsilon, batchsize, m, v_dim, stable_mode != 0);
    });
  } else {
    cuoc::launch(dim3(batchsize), dim3(v_dim), (size_t)m * sizeof(float), [&] {
      curobo::optimization::kernel_lbfgs_step<float, false, -1>(step_vec, rho_buffer, y_buffer, s_buffer, q, x_0, grad_0, grad_q, epsilon,
                                                                batchsize, m, v_dim, stable_mode != 0);
    });
  }
  return 0;
}

// kernel_line_search<float, -1> (one block per problem, opt_dim threads); reference launch: cuda_core_backend/optimization.py:21-110.
extern "C" int ref_line_search(float *best_cost, float *best_action, int16_t *best_iteration, int16_t *current_iteration,
                               uint8_t *converged_global, int convergence_iteration, float cost_delta_threshold,
                               float cost_relative_threshold, float *exploration_cost, float *exploration_action,
                               float *exploration_gradient, int32_t *exploration_idx, float *selected_cost, float *selected_action,
                               float *selected_gradient, int32_t *selected_idx, const float *search_cost, const float *search_action,
                               const float *search_gradient, const float *step_direction, const float *search_magnitudes, float c_1,
                               float c_2, int strong_wolfe, int approx_wolfe, int n_linesearch, int opt_dim, int batchsize) {
  cuoc::launch(dim3(batchsize), dim3(opt_dim), 0, [&] {
    curobo::optimization::line_search::kernel_line_search<float, -1>(
        best_cost, best_action, best_iteration, current_iteration, converged_global, convergence_iteration, cost_delta_threshold,
        cost_relative_threshold, exploration_cost, exploration_action, exploration_gradient, exploration_idx, selected_cost,
        selected_action, selected_gradient, selected_idx, search_cost, search_action, search_gradient, step_direction,
        search_magnitudes, c_1, c_2, strong_wolfe != 0, approx_wolfe != 0, n_linesearch, opt_dim, batchsize);
  });
  return 0;
}

// With one thread per element the kernels return early in threads past the batch, BEFORE the block loads the link constants
// cooperatively -- a partial last block then misses some links.  Every block is therefore full here: the block size is the
// largest divisor of the batch size up to 32.
static int full_block_size(int B) {
  for (int b = std::min(B, 32); b > 1; b--)
    if (B % b == 0) return b;
  return 1;
}

// rnea_forward_kernel<N_LINKS, N_DOF, 1, false> / rnea_backward_kernel<N_LINKS, N_DOF, 1, false> (one thread per element; the
// link and dof counts are template parameters there: franka 13 / 7 and unitree_g1 56 / 49 are instantiated);
// cuda_core_backend/dynamics.py:24-260.  forward_cache is the reference's own [batch, links * 20] scratch.
template <int L, int D>
static int rnea_fwd(float *tau, const float *q, const float *qd, const float *qdd, const float *fixed, const float *masses,
                    const float *inertias, const int8_t *jtype, const int16_t *jmap, const int16_t *lmap, const float *joff,
                    const float *gravity, const int16_t *level_starts, const int16_t *level_links, float *cache, int B, int n_levels) {
  const int bpb = full_block_size(B);
  cuoc::launch(dim3(B / bpb), dim3(bpb), (size_t)1 << 20, [&] {
    curobo::dynamics::rnea_forward_kernel<L, D, 1, false>(tau, q, qd, qdd, fixed, masses, inertias, jtype, jmap, lmap, joff, gravity,
                                                         level_starts, level_links, cache, nullptr, B, n_levels);
  });
  return 0;
}
template <int L, int D>
static int rnea_bwd(float *gq, float *gqd, float *gqdd, const float *gtau, const float *q, const float *qd, const float *fixed,
                    const float *masses, const float *inertias, const int8_t *jtype, const int16_t *jmap, const int16_t *lmap,
                    const float *joff, const float *gravity, const int16_t *level_starts, const int16_t *level_links, const float *cache,
                    int B, int n_levels) {
  const int bpb = full_block_size(B);
  cuoc::launch(dim3(B / bpb), dim3(bpb), (size_t)1 << 20, [&] {
    curobo::dynamics::rnea_backward_kernel<L, D, 1, false>(gq, gqd, gqdd, nullptr, gtau, q, qd, fixed, masses, inertias, jtype, jmap,
                                                          lmap, joff, gravity, level_starts, level_links, cache, B, n_levels);
  });
  return 0;
}
extern "C" int ref_rnea_forward(float *tau, const float *q, const float *qd, const float *qdd, const float *fixed, const float *masses,
                                const float *inertias, const int8_t *jtype, const int16_t *jmap, const int16_t *lmap, const float *joff,
                                const float *gravity, const int16_t *level_starts, const int16_t *level_links, float *cache, int B,
                                int n_levels, int num_links, int num_dof) {
  if (num_links == 13 && num_dof == 7) return rnea_fwd<13, 7>(tau, q, qd, qdd, fixed, masses, inertias, jtype, jmap, lmap, joff, gravity, level_starts, level_links, cache, B, n_levels);
  if (num_links == 56 && num_dof == 49) return rnea_fwd<56, 49>(tau, q, qd, qdd, fixed, masses, inertias, jtype, jmap, lmap, joff, gravity, level_starts, level_links, cache, B, n_levels);
  return 1;
}
extern "C" int ref_rnea_backward(float *gq, float *gqd, float *gqdd, const float *gtau, const float *q, const float *qd, const float *fixed,
                                 const float *masses, const float *inertias, const int8_t *jtype, const int16_t *jmap, const int16_t *lmap,
                                 const float *joff, const float *gravity, const int16_t *level_starts, const int16_t *level_links,
                                 const float *cache, int B, int n_levels, int num_links, int num_dof) {
  if (num_links == 13 && num_dof == 7) return rnea_bwd<13, 7>(gq, gqd, gqdd, gtau, q, qd, fixed, masses, inertias, jtype, jmap, lmap, joff, gravity, level_starts, level_links, cache, B, n_levels);
  if (num_links == 56 && num_dof == 49) return rnea_bwd<56, 49>(gq, gqd, gqdd, gtau, q, qd, fixed, masses, inertias, jtype, jmap, lmap, joff, gravity, level_starts, level_links, cache, B, n_levels);
  return 1;
}

// interpolate_bspline_single_dt_kernel<float, Degree, MATRIX>; cuda_core_backend/trajectory.py:207-301 (256 threads per block)
template <int DEG>
static void bspline_single_dt(float *p, float *v, float *a, float *j, float *out_dt, const float *u, const float *knot_dt, const float *sp,
                              const float *sv, const float *sa, const float *sj, const float *gp, const float *gv, const float *ga,
                              const float *gj, const int32_t *start_idx, const int32_t *goal_idx, const float *interp_dt,
                              const uint8_t *implicit, const int32_t *interp_horizon, int B, int max_out, int D, int K) {
  const int k_size = B * D * max_out, threads = std::min(k_size, 256);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    bs::interpolate_bspline_single_dt_kernel<float, DEG, bs::BasisBackend::MATRIX>(p, v, a, j, out_dt, u, knot_dt, sp, sv, sa, sj, gp, gv,
                                                                                  ga, gj, start_idx, goal_idx, interp_dt, implicit,
                                                                                  interp_horizon, B, max_out, D, K);
  });
}
extern "C" int ref_bspline_single_dt(float *p, float *v, float *a, float *j, float *out_dt, const float *u, const float *knot_dt,
                                     const float *sp, const float *sv, const float *sa, const float *sj, const float *gp, const float *gv,
                                     const float *ga, const float *gj, const int32_t *start_idx, const int32_t *goal_idx,
                                     const float *interp_dt, const uint8_t *implicit, const int32_t *interp_horizon, int B, int max_out,
                                     int D, int K, int degree) {
#define CUOC_SDT(DEG) bspline_single_dt<DEG>(p, v, a, j, out_dt, u, knot_dt, sp, sv, sa, sj, gp, gv, ga, gj, start_idx, goal_idx, interp_dt, implicit, interp_horizon, B, max_out, D, K)
  if (degree == 3) CUOC_SDT(3);
  else if (degree == 4) CUOC_SDT(4);
  else if (degree == 5) CUOC_SDT(5);
  else return 1;
#undef CUOC_SDT
  return 0;
}

// legacy POSITION control space: position_clique_loop_idx_fwd_kernel<float, true> / ..._bwd_kernel<float, true> (five-point
// stencils); cuda_core_backend/trajectory.py:304-465 (128 threads per block)
extern "C" int ref_differentiation_position_forward(float *p, float *v, float *a, float *j, float *out_dt, const float *u, const float *sp,
                                                    const float *sv, const float *sa, const float *gp, const float *gv, const float *ga,
                                                    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,
                                                    const uint8_t *implicit, int B, int H, int D) {
  const int k_size = B * D * H, threads = std::min(k_size, 128);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    curobo::trajectory::legacy::position_clique_loop_idx_fwd_kernel<float, true>(p, v, a, j, out_dt, u, sp, sv, sa, gp, gv, ga, start_idx,
                                                                                goal_idx, traj_dt, implicit, B, H, D);
  });
  return 0;
}
extern "C" int ref_differentiation_position_backward(float *out, const float *gp, const float *gv, const float *ga, const float *gj,
                                                     const float *traj_dt, const int32_t *dt_idx, const uint8_t *implicit, int B, int H,
                                                     int D) {
  const int k_size = B * D * (H - 4), threads = std::min(k_size, 128);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    curobo::trajectory::legacy::position_clique_loop_idx_bwd_kernel<float, true>(out, gp, gv, ga, gj, traj_dt, dt_idx, implicit, B, H, D);
  });
  return 0;
}

// legacy ACCELERATION control space: acceleration_loop_idx_kernel<float, H> / acceleration_loop_idx_rk2_kernel<float, H> (the horizon
// is a template parameter there; 32 and 64 are instantiated, the call takes the smallest that holds `horizon`);
// cuda_core_backend/trajectory.py:468-544
template <int MAXH>
static void accel_launch(bool rk2, float *p, float *v, float *a, float *j, const float *u, const float *sp, const float *sv, const float *sa,
                         const int32_t *start_idx, const float *traj_dt, int B, int H, int D) {
  const int k_size = B * D, threads = std::min(k_size, 512);
  cuoc::launch(dim3((k_size + threads - 1) / threads), dim3(threads), 0, [&] {
    if (rk2) curobo::trajectory::legacy::acceleration_loop_idx_rk2_kernel<float, MAXH>(p, v, a, j, u, sp, sv, sa, start_idx, traj_dt, B, H, D);
    else curobo::trajectory::legacy::acceleration_loop_idx_kernel<float, MAXH>(p, v, a, j, u, sp, sv, sa, start_idx, traj_dt, B, H, D);
  });
}
extern "C" int ref_integration_acceleration(float *p, float *v, float *a, float *j, const float *u, const float *sp, const float *sv,
                                            const float *sa, const int32_t *start_idx, const float *traj_dt, int B, int H, int D,
                                            int use_rk2) {
  if (H <= 32) accel_launch<32>(use_rk2 != 0, p, v, a, j, u, sp, sv, sa, start_idx, traj_dt, B, H, D);
  else if (H <= 64) accel_launch<64>(use_rk2 != 0, p, v, a, j, u, sp, sv, sa, start_idx, traj_dt, B, H, D);
  else return 1;
  return 0;
}
