// cost.hip -- goal-set tool-pose distance and c-space POSITION bound cost.
// Reference (NVIDIA Warp, no backend hook): cost/wp_tool_pose.py:61-692 (ToolPoseDistance :698),
// cost/wp_cspace_position.py:232-362, cost/warp_bound_util.py:9-100.
//
// Both are tiny elementwise maps (one lane per (batch, horizon, link) resp. (batch, horizon,
// dof)); they exist as stand-alone entry points for the drop-in API and are also inlined into
// the fused IK rollout kernel (rollout_fused.hip) where launch latency matters.
#include "cost_device.hpp"

namespace curobo_hip {

__global__ void __launch_bounds__(256) tool_pose_distance_kernel(const ToolPoseArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.batch * a.horizon * a.num_links;
  if (tid >= total) return;
  const int hl = a.horizon * a.num_links;
  const int b = (int)(tid / hl);
  const int h = (int)((tid - (long)b * hl) / a.num_links);
  const int l = (int)(tid - (long)b * hl - (long)h * a.num_links);
  const float *cp = a.current_position + tid * 3;
  const float4 cqw = reinterpret_cast<const float4 *>(a.current_quat)[tid];
  const ToolPoseResult r = tool_pose_distance_point(a, b, h, l, make_f3(cp[0], cp[1], cp[2]), cqw);
  a.out_distance[2 * tid] = r.position_cost;
  a.out_distance[2 * tid + 1] = r.rotation_cost;
  a.out_goalset_idx[tid] = r.goalset_idx;
  a.out_position_distance[tid] = r.position_distance;
  a.out_rotation_distance[tid] = r.rotation_distance;
  float *pg = a.out_position_gradient + tid * 3;
  pg[0] = r.position_gradient.x; pg[1] = r.position_gradient.y; pg[2] = r.position_gradient.z;
  reinterpret_cast<float4 *>(a.out_rotation_gradient)[tid] = r.quat_rate_wxyz;
}

__global__ void __launch_bounds__(256) cspace_position_kernel(const CspacePosArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.batch * a.horizon * a.dof;
  if (tid >= total) return;
  const int b = (int)(tid / ((long)a.horizon * a.dof));
  const int d = (int)(tid % a.dof);
  float gp, gt;
  const float c = cspace_position_point(a, b, d, a.pos[tid], a.effort ? a.effort[tid] : 0.0f, gp, gt);
  a.out_cost[tid] = c;
  if (a.write_grad) {
    a.out_grad_p[tid] = gp;
    if (a.out_grad_tau) a.out_grad_tau[tid] = gt;
  }
}


// Per-row cost aggregation for teleport (horizon 1) rollouts such as IK: one 16-lane DPP row per
// rollout row sums pose (2 per link), c-space (per dof), self-collision and scene (per sphere)
// costs, and folds the c-space gradient into grad_q (the reference does this with torch cat/sum
// and autograd accumulation: rollout/metrics.py:233-265).
__global__ void __launch_bounds__(256) rollout_point_aggregate_kernel(
    float *out_cost, float *grad_q, const float *pose_cost, const float *cspace_cost, const float *cspace_grad,
    const float *self_cost, const float *scene_cost, int rows, int num_links, int dof, int nspheres) {
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) / 16, lane = threadIdx.x % 16;
  const bool valid = grp < rows;
  float acc = 0.0f;
  if (valid) {
    if (pose_cost)
      for (int i = lane; i < 2 * num_links; i += 16) acc += pose_cost[(size_t)grp * 2 * num_links + i];
    if (cspace_cost)
      for (int i = lane; i < dof; i += 16) acc += cspace_cost[(size_t)grp * dof + i];
    if (scene_cost)
      for (int i = lane; i < nspheres; i += 16) acc += scene_cost[(size_t)grp * nspheres + i];
    if (self_cost && lane == 0) acc += self_cost[grp];
    if (cspace_grad && grad_q)
      for (int i = lane; i < dof; i += 16) grad_q[(size_t)grp * dof + i] += cspace_grad[(size_t)grp * dof + i];
  }
  acc = row16_sum(acc);
  if (valid && lane == 0) out_cost[grp] = acc;
}

__global__ void __launch_bounds__(256) cspace_state_kernel(const CspaceStateArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.batch * a.horizon * a.dof;
  if (tid >= total) return;
  const int b = (int)(tid / ((long)a.horizon * a.dof));
  const int h = (int)((tid - (long)b * a.horizon * a.dof) / a.dof);
  const int d = (int)(tid % a.dof);
  const float x[5] = {a.pos[tid], a.vel[tid], a.acc[tid], a.jerk[tid], a.effort ? a.effort[tid] : 0.0f};
  float g[5];
  a.out_cost[tid] = cspace_state_point(a, b, h, d, x, g);
  if (a.write_grad) {
    a.out_gp[tid] = g[0]; a.out_gv[tid] = g[1]; a.out_ga[tid] = g[2]; a.out_gj[tid] = g[3];
    if (a.out_gtau) a.out_gtau[tid] = g[4];
  }
}

// c-space L2 distance to a target configuration (reference forward_l2_warp,
// cost/wp_torch_cspace_dist.py:12-78): cost = w * r_d * (q - target)^2 per (batch, horizon, dof) with the
// per-dof weight r_d taken from the terminal / non-terminal table; entries of zero weight are NOT
// written (the reference returns before its stores), so the caller's buffers keep their content there.
struct CspaceL2Args {
  float *out_cost, *out_gp;
  const float *pos, *target, *weight, *terminal_dof_weight, *non_terminal_dof_weight;
  const int32_t *target_idx;
  int write_grad, batch, horizon, dof;
};

__global__ void __launch_bounds__(256) cspace_l2_kernel(const CspaceL2Args a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.batch * a.horizon * a.dof;
  if (tid >= total) return;
  const int b = (int)(tid / ((long)a.horizon * a.dof));
  const int h = (int)((tid - (long)b * a.horizon * a.dof) / a.dof);
  const int d = (int)(tid % a.dof);
  const float w = (h < a.horizon - 1 ? a.non_terminal_dof_weight[d] : a.terminal_dof_weight[d]) * a.weight[0];
  if (w == 0.0f) return;
  const float err = a.pos[tid] - a.target[(size_t)a.target_idx[b] * a.dof + d];
  a.out_cost[tid] = w * err * err;
  if (a.write_grad) a.out_gp[tid] = 2.0f * w * err;
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_tool_pose_distance(
    float *out_distance, float *out_position_distance, float *out_rotation_distance,
    float *out_position_gradient, float *out_rotation_gradient, int32_t *out_goalset_idx,
    const float *current_position, const float *current_quat, const float *goal_position,
    const float *goal_quat, const int32_t *idxs_goal, const float *position_orientation_weight,
    const float *terminal_pose_axes_weight_factor, const float *non_terminal_pose_axes_weight_factor,
    const float *terminal_pose_convergence_tolerance, const float *non_terminal_pose_convergence_tolerance,
    const uint8_t *project_distance_to_goal, int batch_size, int horizon, int num_links, int num_goalset,
    int rotation_method, curobo_hip_stream_t stream) {
  const char *what = "tool_pose_distance";
  CUROBO_REQUIRE(rotation_method >= 0 && rotation_method <= 2, "%s: rotation_method must be 0, 1 or 2", what);
  CUROBO_REQUIRE(num_goalset >= 1 && num_links >= 1 && horizon >= 1, "%s: bad dimensions", what);
  CUROBO_REQUIRE(((uintptr_t)current_quat & 15) == 0 && ((uintptr_t)out_rotation_gradient & 15) == 0,
                 "%s: quaternion buffers must be 16-byte aligned", what);
  const long total = (long)batch_size * horizon * num_links;
  if (total == 0) return CUROBO_HIP_OK;
  ToolPoseArgs a{out_distance, out_position_distance, out_rotation_distance, out_position_gradient,
                 out_rotation_gradient, out_goalset_idx, current_position, current_quat, goal_position, goal_quat,
                 idxs_goal, position_orientation_weight, terminal_pose_axes_weight_factor,
                 non_terminal_pose_axes_weight_factor, terminal_pose_convergence_tolerance,
                 non_terminal_pose_convergence_tolerance, project_distance_to_goal, batch_size, horizon, num_links,
                 num_goalset, rotation_method};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tool_pose_distance_kernel, dim3((unsigned)ceil_div_l(total, 256)), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_cspace_position_cost(
    float *out_cost, float *out_grad_p, float *out_grad_tau, const float *pos, const float *effort,
    const float *cspace_target, const int32_t *cspace_target_idx, const float *p_b, const float *effort_b,
    const float *weight, const float *activation_distance, const float *cspace_target_weight,
    const float *cspace_target_dof_weight, const float *squared_l2_reg_weight, const float *current_position,
    const float *current_velocity, const int32_t *idxs_current_state, const float *v_b, const float *state_dt,
    int write_grad, int batch_size, int horizon, int dof, curobo_hip_stream_t stream) {
  const char *what = "cspace_position_cost";
  CUROBO_REQUIRE(dof >= 1 && horizon >= 1, "%s: bad dimensions", what);
  const long total = (long)batch_size * horizon * dof;
  if (total == 0) return CUROBO_HIP_OK;
  CspacePosArgs a{out_cost, out_grad_p, out_grad_tau, pos, effort, cspace_target, cspace_target_idx, p_b, effort_b,
                  weight, activation_distance, cspace_target_weight, cspace_target_dof_weight, squared_l2_reg_weight,
                  current_position, current_velocity, idxs_current_state, v_b, state_dt, write_grad, batch_size,
                  horizon, dof};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cspace_position_kernel, dim3((unsigned)ceil_div_l(total, 256)), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_rollout_point_aggregate(float *out_cost, float *grad_q, const float *pose_cost,
                                                     const float *cspace_cost, const float *cspace_grad,
                                                     const float *self_cost, const float *scene_cost, int rows,
                                                     int num_links, int dof, int num_spheres,
                                                     curobo_hip_stream_t stream) {
  const char *what = "rollout_point_aggregate";
  CUROBO_REQUIRE(rows >= 0 && num_links >= 0 && dof >= 1 && num_spheres >= 0, "%s: bad dimensions", what);
  if (rows == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rollout_point_aggregate_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, st, out_cost,
                     grad_q, pose_cost, cspace_cost, cspace_grad, self_cost, scene_cost, rows, num_links, dof,
                     num_spheres);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_cspace_state_cost(
    float *out_cost, float *out_grad_p, float *out_grad_v, float *out_grad_a, float *out_grad_j, float *out_grad_tau,
    const float *pos, const float *vel, const float *acc, const float *jerk, const float *effort, const float *state_dt,
    const float *target_joint_position, const int32_t *idxs_target_joint_position, const float *p_b, const float *v_b,
    const float *a_b, const float *j_b, const float *effort_b, const float *weight, const float *activation_distance,
    const float *squared_l2_regularization_weights, const float *cspace_target_weight,
    const float *cspace_non_terminal_weight_factor, const float *cspace_target_dof_weight, int write_grad,
    int batch_size, int horizon, int dof, int retime_weights, int retime_regularization_weights,
    curobo_hip_stream_t stream) {
  const char *what = "cspace_state_cost";
  CUROBO_REQUIRE(horizon >= 1 && dof >= 1, "%s: bad dimensions", what);
  CUROBO_REQUIRE(!write_grad || (out_grad_p && out_grad_v && out_grad_a && out_grad_j), "%s: gradient buffers are NULL", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  CspaceStateArgs a{out_cost, out_grad_p, out_grad_v, out_grad_a, out_grad_j, out_grad_tau, pos, vel, acc, jerk, effort,
                    state_dt, target_joint_position, idxs_target_joint_position, p_b, v_b, a_b, j_b, effort_b, weight,
                    activation_distance, squared_l2_regularization_weights, cspace_target_weight,
                    cspace_non_terminal_weight_factor, cspace_target_dof_weight, write_grad, batch_size, horizon, dof,
                    retime_weights, retime_regularization_weights};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cspace_state_kernel, dim3((unsigned)ceil_div_l((long)batch_size * horizon * dof, 256)), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_cspace_l2_distance(float *out_cost, float *out_grad_p, const float *pos, const float *target,
                                                const int32_t *target_idx, const float *weight,
                                                const float *terminal_dof_weight, const float *non_terminal_dof_weight,
                                                int write_grad, int batch_size, int horizon, int dof,
                                                curobo_hip_stream_t stream) {
  const char *what = "cspace_l2_distance";
  CUROBO_REQUIRE(batch_size >= 0 && horizon >= 1 && dof >= 1, "%s: bad sizes", what);
  CUROBO_REQUIRE(out_cost && pos && target && target_idx && weight && terminal_dof_weight && non_terminal_dof_weight,
                 "%s: NULL argument", what);
  CUROBO_REQUIRE(!write_grad || out_grad_p, "%s: write_grad without a gradient buffer", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  CspaceL2Args a{out_cost, out_grad_p, pos, target, weight, terminal_dof_weight, non_terminal_dof_weight, target_idx,
                 write_grad, batch_size, horizon, dof};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cspace_l2_kernel, dim3((unsigned)ceil_div_l((long)batch_size * horizon * dof, 256)), dim3(256), 0, st, a);
  return check_launch(what, st);
}
