// seed_ik.hip -- the iteration-state update of the Levenberg-Marquardt seed-IK solver in one launch.
//
// Reference (all torch, ~25 elementwise launches per iteration under a CUDA graph):
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:338-387  joint-limit residual rows
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:292-305  pose-error reduction
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:464-495  combination of the residual blocks
//   curobo/_src/solver/seed_ik/seed_iteration_state_manager.py:74-260  trust ratio, step acceptance,
//                                                                     damping update, state selection,
//                                                                     convergence flags
// One 16-lane DPP row per problem (4 problems per wavefront): lanes stride over the dofs / Jacobian
// elements, row reductions for the sums.  Everything is a pure function of the candidate buffers
// and the previous state, so the kernel is graph-capturable and deterministic.
#include "common.hpp"
#include "self_device.hpp"

namespace curobo_hip {

struct SeedIkUpdateArgs {
  // state (read-modify-write)
  float *q, *jacobian, *jTerror, *error_norm, *position_error, *orientation_error, *lambda_damping;
  uint8_t *success, *improvement;
  // candidate
  const float *cand_q, *cand_pose_jacobian, *cand_pose_jTerror, *cand_pose_cost, *cand_position_distance,
      *cand_rotation_distance, *pred_reduction;
  const float *action_min, *action_max;
  // optional velocity clamping of the limits (seed_ik_error_calculator.py:355-363)
  const float *current_position, *dt, *velocity_limits;
  // optional velocity / acceleration regularisation rows (seed_ik_error_calculator.py:389-456): with
  // v = (q - current_position) / dt,  r_v = sqrt(w_v dt) v  and  r_a = sqrt(w_a) (v - current_velocity); both blocks are
  // DIAGONAL in q, like the joint-limit block, and the LM step only sees J^T J and J^T r, so the three diagonal rows of a
  // dof are stored as ONE row of magnitude sqrt(sum of squares) (same normal equations, same predicted reduction)
  const float *current_velocity;
  float velocity_weight, acceleration_weight;
  float joint_limit_weight, rho_min, lambda_factor, lambda_min, lambda_max, conv_pos_tol, conv_ori_tol, conv_jl_weight;
  int n, D, T, initial;
};

constexpr int kRow = 16;

// velocity / acceleration residuals of dof d: squared Jacobian diagonal, J^T r and squared error
__device__ __forceinline__ void vel_acc_rows(const SeedIkUpdateArgs &a, int p, int d, float x, float &diag2, float &jtr, float &err2) {
  diag2 = jtr = err2 = 0.0f;
  if (a.current_position == nullptr || a.dt == nullptr || !(a.dt[p] > 0.0f)) return;
  if (a.velocity_weight <= 0.0f && a.acceleration_weight <= 0.0f) return;
  const float dt = fmaxf(a.dt[p], 1e-10f), inv_dt = 1.0f / dt;
  const float v = (x - a.current_position[(size_t)p * a.D + d]) * inv_dt;
  if (a.velocity_weight > 0.0f) {
    const float sw = sqrtf(a.velocity_weight * dt), e = sw * v, jd = sw * inv_dt;
    diag2 += jd * jd; jtr += jd * e; err2 += e * e;
  }
  if (a.acceleration_weight > 0.0f && a.current_velocity != nullptr) {
    const float sw = sqrtf(a.acceleration_weight), e = sw * (v - a.current_velocity[(size_t)p * a.D + d]), jd = sw * inv_dt;
    diag2 += jd * jd; jtr += jd * e; err2 += e * e;
  }
}

__device__ __forceinline__ float row16_maxf(float v) { return row16_max(v); }

__global__ void __launch_bounds__(256) seed_ik_update_kernel(const SeedIkUpdateArgs a) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) / kRow, lane = threadIdx.x % kRow;
  // rows beyond n keep running (DPP row reductions need all lanes of a wave), with clamped loads and no stores
  const bool live = row < a.n;
  const int p = live ? row : a.n - 1;
  const int D = a.D, T = a.T, R = 6 * T + D;
  const float *cq = a.cand_q + (size_t)p * D;

  // ---- joint-limit residual of the candidate: error, diagonal Jacobian, J^T e contribution
  float jl_sum = 0.0f;
  bool inside = true;  // strictly inside [action_min, action_max] (convergence check on the SELECTED q, below)
  for (int d = lane; d < D; d += kRow) {
    float lo = a.action_min[d], hi = a.action_max[d];
    if (a.current_position && a.velocity_limits) {
      const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
      lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
      hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
    }
    const float x = cq[d];
    const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
    jl_sum += a.joint_limit_weight * (lv + uv);
    float d2, jt, e2;
    vel_acc_rows(a, p, d, x, d2, jt, e2);
    jl_sum += e2;  // error norms of the velocity / acceleration blocks (_combine_errors, :464-495)
  }
  jl_sum = row16_sum(jl_sum);

  // ---- candidate error norm = sum of the pose cost terms + joint-limit errors; worst tool frame errors
  float pose_sum = 0.0f, pos_e = 0.0f, ori_e = 0.0f;
  for (int i = lane; i < 2 * T; i += kRow) pose_sum += a.cand_pose_cost[(size_t)p * 2 * T + i];
  for (int t = lane; t < T; t += kRow) {
    pos_e = fmaxf(pos_e, a.cand_position_distance[(size_t)p * T + t]);
    ori_e = fmaxf(ori_e, a.cand_rotation_distance[(size_t)p * T + t]);
  }
  pose_sum = row16_sum(pose_sum);
  pos_e = row16_maxf(pos_e);
  ori_e = row16_maxf(ori_e);
  const float cand_norm = pose_sum + jl_sum;

  // ---- trust-region ratio, acceptance, damping (seed_iteration_state_manager.py:124-180)
  bool accepted = true;
  float lambda = a.lambda_damping[p];
  if (!a.initial) {
    const float actual = a.error_norm[p] - cand_norm;
    const float rho = actual / (a.pred_reduction[p] + 1e-8f);
    accepted = rho >= a.rho_min;  // false for NaN
    lambda = accepted ? lambda / a.lambda_factor : lambda * a.lambda_factor;
    lambda = fminf(fmaxf(lambda, a.lambda_min), a.lambda_max);
  }

  // ---- state selection: accepted -> candidate values, rejected -> keep (error_norm is ALWAYS the
  // candidate's, seed_iteration_state_manager.py:117)
  if (accepted) {
    float *J = a.jacobian + (size_t)p * R * D;
    const float *cJ = a.cand_pose_jacobian + (size_t)p * 6 * T * D;
    if (live) {
      for (int i = lane; i < 6 * T * D; i += kRow) J[i] = cJ[i];
      for (int i = lane; i < D * D; i += kRow) J[6 * T * D + i] = 0.0f;
    }
    for (int d = lane; d < D; d += kRow) {
      float lo = a.action_min[d], hi = a.action_max[d];
      if (a.current_position && a.velocity_limits) {
        const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
        lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
        hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
      }
      const float x = cq[d];
      const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
      const float err = a.joint_limit_weight * (lv + uv);
      const float diag = a.joint_limit_weight * ((lv > 0.0f ? -1.0f : 0.0f) + (uv > 0.0f ? 1.0f : 0.0f));
      float d2, jt, e2;
      vel_acc_rows(a, p, d, x, d2, jt, e2);
      if (live) {
        a.q[(size_t)p * D + d] = x;
        a.jTerror[(size_t)p * D + d] = a.cand_pose_jTerror[(size_t)p * D + d] + diag * err + jt;
      }
    }
    // the zero fill above and the diagonal below touch the same elements from different lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < D; d += kRow) {
      float lo = a.action_min[d], hi = a.action_max[d];
      if (a.current_position && a.velocity_limits) {
        const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
        lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
        hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
      }
      const float x = cq[d];
      const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
      const float diag = a.joint_limit_weight * ((lv > 0.0f ? -1.0f : 0.0f) + (uv > 0.0f ? 1.0f : 0.0f));
      float d2, jt, e2;
      vel_acc_rows(a, p, d, x, d2, jt, e2);
      if (live) J[(size_t)(6 * T + d) * D + d] = d2 > 0.0f ? sqrtf(diag * diag + d2) : diag;
    }
  } else {
    pos_e = a.position_error[p];
    ori_e = a.orientation_error[p];
  }

  // ---- convergence flag on the selected state (:222-260)
  const float *sq = accepted ? cq : a.q + (size_t)p * D;
  for (int d = lane; d < D; d += kRow) {
    const float x = sq[d];
    inside = inside && (x > a.action_min[d]) && (x < a.action_max[d]);
  }
  const float outside = row16_maxf(inside ? 0.0f : 1.0f);
  bool ok = pos_e < a.conv_pos_tol && ori_e < a.conv_ori_tol;
  if (a.conv_jl_weight > 0.0f) ok = ok && outside == 0.0f;
  if (live && lane == 0) {
    a.error_norm[p] = cand_norm;
    a.position_error[p] = pos_e;
    a.orientation_error[p] = ori_e;
    a.lambda_damping[p] = lambda;
    a.success[p] = ok ? 1 : 0;
    a.improvement[p] = accepted ? 1 : 0;
  }
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_seed_ik_update_state(
    float *joint_position, float *jacobian, float *jTerror, float *error_norm, float *position_error,
    float *orientation_error, float *lambda_damping, uint8_t *success, uint8_t *improvement,
    const float *candidate_joint_position, const float *candidate_pose_jacobian, const float *candidate_pose_jTerror,
    const float *candidate_pose_cost, const float *candidate_position_distance, const float *candidate_rotation_distance,
    const float *predicted_reduction, const float *action_min, const float *action_max, const float *current_position,
    const float *dt, const float *velocity_limits, const float *current_velocity, float velocity_weight,
    float acceleration_weight, float joint_limit_weight, float rho_min, float lambda_factor,
    float lambda_min, float lambda_max, float convergence_position_tolerance, float convergence_orientation_tolerance,
    float convergence_joint_limit_weight, int num_problems, int dof, int num_tool_frames, int initial,
    curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(num_problems >= 0 && dof >= 1 && num_tool_frames >= 1, "seed_ik_update_state: bad sizes (n=%d, dof=%d, T=%d)",
                 num_problems, dof, num_tool_frames);
  CUROBO_REQUIRE(initial || predicted_reduction, "seed_ik_update_state: predicted_reduction is NULL%s", "");
  CUROBO_REQUIRE(!current_position || dt, "seed_ik_update_state: current_position needs dt%s", "");
  if (num_problems == 0) return CUROBO_HIP_OK;
  SeedIkUpdateArgs a;
  a.q = joint_position; a.jacobian = jacobian; a.jTerror = jTerror; a.error_norm = error_norm;
  a.position_error = position_error; a.orientation_error = orientation_error; a.lambda_damping = lambda_damping;
  a.success = success; a.improvement = improvement;
  a.cand_q = candidate_joint_position; a.cand_pose_jacobian = candidate_pose_jacobian; a.cand_pose_jTerror = candidate_pose_jTerror;
  a.cand_pose_cost = candidate_pose_cost; a.cand_position_distance = candidate_position_distance;
  a.cand_rotation_distance = candidate_rotation_distance; a.pred_reduction = predicted_reduction;
  a.action_min = action_min; a.action_max = action_max; a.current_position = current_position; a.dt = dt;
  a.velocity_limits = velocity_limits;
  a.current_velocity = current_velocity; a.velocity_weight = velocity_weight; a.acceleration_weight = acceleration_weight;
  a.joint_limit_weight = joint_limit_weight; a.rho_min = rho_min; a.lambda_factor = lambda_factor; a.lambda_min = lambda_min;
  a.lambda_max = lambda_max; a.conv_pos_tol = convergence_position_tolerance; a.conv_ori_tol = convergence_orientation_tolerance;
  a.conv_jl_weight = convergence_joint_limit_weight;
  a.n = num_problems; a.D = dof; a.T = num_tool_frames; a.initial = initial;
  hipStream_t st = (hipStream_t)stream;
  const int rows_per_block = 256 / kRow;
  hipLaunchKernelGGL(seed_ik_update_kernel, dim3(ceil_div(num_problems, rows_per_block)), dim3(256), 0, st, a);
  return check_launch("seed_ik_update_state", st);
}
