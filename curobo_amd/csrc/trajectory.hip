// trajectory.hip -- uniform B-spline knots -> (position, velocity, acceleration, jerk) and the
// VJP back to the knots.  Reference: kernels/trajectory/bspline/bspline_kernel.cuh:81-151,332-380,
// bspline_interpolation.cuh:95-297, bspline_boundary_constraint.cuh:52-367,
// basis/bspline_basis_matrix.cuh:20-156 (MATRIX basis, the backend both reference launchers
// select), bspline_context.cuh:73-170, bspline_gradient_util.cuh:141-227.
//
// gfx950 design: both directions are tiny, purely bandwidth/latency-bound elementwise maps
// (~1 KB per trajectory).  Forward: one lane per (b, h, d) with d fastest, so the four output
// streams are written fully coalesced.  Backward: one lane per (b, knot, d) that loops over the
// `interpolation_steps` samples itself -- the reference spreads those over lanes and needs a
// segmented warp-32 shuffle reduction (bspline_common.cuh:60-82) that has no wave64 analogue
// worth keeping; a 2-4 iteration in-register loop is cheaper than any cross-lane traffic.
#include "bspline_device.hpp"

namespace curobo_hip {

template <int DEG>
__global__ void __launch_bounds__(256) bspline_forward_kernel(const BsFwdArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dof = a.dof, ph = a.padded_horizon;
  const int b = (int)(tid / ((long)dof * ph));
  if (b >= a.batch) return;
  const int d = (int)(tid % dof);
  const int h = (int)((tid / dof) % ph);
  float o[4];
  const float interpolated_dt = bspline_sample<DEG>(a, b, h, d, o);
  const size_t addr = ((size_t)b * ph + h) * dof + d;
  a.out_pos[addr] = o[0]; a.out_vel[addr] = o[1]; a.out_acc[addr] = o[2]; a.out_jerk[addr] = o[3];
  if (h == 0 && d == 0) a.out_dt[b] = interpolated_dt;
}

// interpolate_bspline_single_dt_kernel (bspline_kernel.cuh:221-270): one dt for every trajectory,
// a per-trajectory horizon; points past a trajectory's horizon repeat its last sample (the
// reference's clamp of the knot index), output stride = max_out_tsteps
template <int DEG>
__global__ void __launch_bounds__(256) bspline_single_dt_kernel(const BsFwdArgs a, const float *interpolation_dt,
                                                                const int32_t *interpolation_horizon, int max_out) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dof = a.dof;
  const int b = (int)(tid / ((long)dof * max_out));
  if (b >= a.batch) return;
  const int d = (int)(tid % dof);
  const int h = (int)((tid / dof) % max_out);
  const int new_horizon = min(interpolation_horizon[b], max_out - 1);
  float o[4];
  const float dt = bspline_sample_at<DEG>(a, b, h, d, new_horizon + 1, interpolation_dt[0], o);
  const size_t addr = ((size_t)b * max_out + h) * dof + d;
  a.out_pos[addr] = o[0]; a.out_vel[addr] = o[1]; a.out_acc[addr] = o[2]; a.out_jerk[addr] = o[3];
  if (h == 0 && d == 0) a.out_dt[b] = dt;
}

template <int DEG>
__global__ void __launch_bounds__(256) bspline_backward_kernel(const BsBwdArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dof = a.dof, nk = a.n_knots, ph = a.padded_horizon;
  const int b = (int)(tid / ((long)dof * nk));
  if (b >= a.batch) return;
  const int d = (int)(tid % dof);
  const int k = (int)((tid / dof) % nk);
  const int dto = a.dt_idx[b];
  a.out_grad[((size_t)b * nk + k) * dof + d] = bspline_knot_grad<DEG>(
      a.gin, (size_t)b * ph * dof + d, dof, k, nk, ph, a.traj_dt[dto], a.use_implicit_goal[dto] != 0);
}


// ------------------------------------------------------------------------------------------
// Legacy transitions of the reference's POSITION / ACCELERATION control spaces.
//
// Position clique (legacy/differentiation_position_kernel.cuh:15-232, use_stencil = true as the
// launcher fixes it): the reference's 13-way branch table is a 5-point window sliding over ONE
// extended position sequence
//   P = [e(-3), e(-2), e(-1), x0, u_0 .. u_{A-1}, u_{A-1} x4],   A = horizon - 4,
// e(.) = constant-acceleration back-extrapolation of the start state, u_{A-1} replaced by the goal
// position under use_implicit_goal_state; window of point h = P[h .. h+4].  Written that way here
// (branch-free index clamp instead of the table; identical for horizon >= 9).
__device__ __forceinline__ float clique_P(int j, const float *__restrict__ u, int A, int dof, int d, float x0, float v0,
                                          float a0, float dt, bool use_goal, float goal) {
  const float fixed_jerk = 0.0f;
  if (j == 0) return (3.0f / 2) * (-1 * a0 * (dt * dt) - (dt * dt * dt) * fixed_jerk) - 3.0f * dt * v0 + x0;
  if (j == 1) return -2.0f * a0 * dt * dt - (4.0f / 3) * dt * dt * dt * fixed_jerk - 2.0f * dt * v0 + x0;
  if (j == 2) return -(3.0f / 2) * a0 * dt * dt - (7.0f / 6) * dt * dt * dt * fixed_jerk - dt * v0 + x0;
  if (j == 3) return x0;
  const int i = j - 4;
  if (i >= A - 1) return use_goal ? goal : u[(size_t)(A - 1) * dof + d];
  return u[(size_t)i * dof + d];
}

__global__ void __launch_bounds__(256) differentiation_position_forward_kernel(
    float *out_pos, float *out_vel, float *out_acc, float *out_jerk, float *out_dt, const float *u_position,
    const float *start_pos, const float *start_vel, const float *start_acc, const float *goal_pos,
    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt, const uint8_t *use_implicit_goal,
    int batch, int horizon, int dof) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = (int)(tid / ((long)dof * horizon));
  if (b >= batch) return;
  const int d = (int)(tid % dof);
  const int h = (int)((tid / dof) % horizon);
  const int bo = start_idx[b], go = goal_idx[b];
  const float dt = traj_dt[go], dt_inv = 1.0f / dt;
  const bool use_goal = use_implicit_goal[go] != 0;
  const int A = horizon - 4;
  const float *u = u_position + (size_t)b * A * dof;
  const float x0 = start_pos[bo * dof + d], v0 = start_vel[bo * dof + d], a0 = start_acc[bo * dof + d];
  const float goal = use_goal ? goal_pos[go * dof + d] : 0.0f;
  float p[5];
#pragma unroll
  for (int i = 0; i < 5; i++) p[i] = clique_P(h + i, u, A, dof, d, x0, v0, a0, dt, use_goal, goal);
  const size_t a = ((size_t)b * horizon + h) * dof + d;
  out_pos[a] = p[2];
  out_vel[a] = ((0.083333333f) * p[0] - (0.666666667f) * p[1] + (0.666666667f) * p[3] + (-0.083333333f) * p[4]) * dt_inv;
  out_acc[a] = ((-0.083333333f) * p[0] + (1.333333333f) * p[1] + (-2.5f) * p[2] + (1.333333333f) * p[3] + (-0.083333333f) * p[4]) *
               dt_inv * dt_inv;
  out_jerk[a] = ((-(1.0f / 2.0f)) * p[0] + p[1] - p[3] + ((1.0f / 2.0f)) * p[4]) * (dt_inv * dt_inv * dt_inv);
  if (h == 0 && d == 0) out_dt[b] = dt;
}

// position_clique_loop_idx_bwd_kernel (differentiation_position_kernel.cuh:234-370, stencil variant);
// dof is the fastest thread index here (coalesced), the reference's is the action step
__global__ void __launch_bounds__(256) differentiation_position_backward_kernel(
    float *out_grad, const float *grad_pos, const float *grad_vel, const float *grad_acc, const float *grad_jerk,
    const float *traj_dt, const int32_t *dt_idx, const uint8_t *use_implicit_goal, int batch, int horizon, int dof) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int A = horizon - 4;
  const int b = (int)(tid / ((long)dof * A));
  if (b >= batch) return;
  const int d = (int)(tid % dof);
  const int ah = (int)((tid / dof) % A);
  const int dto = dt_idx[b];
  const float dt_inv = 1.0f / traj_dt[dto];
  const bool use_goal = use_implicit_goal[dto] != 0;
  const float i1 = dt_inv, i2 = dt_inv * dt_inv, i3 = dt_inv * dt_inv * dt_inv;
  const size_t base = (size_t)b * horizon * dof + d;
  float gv[5], ga[5], gj[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    gv[i] = grad_vel[base + (size_t)(ah + i) * dof];
    ga[i] = grad_acc[base + (size_t)(ah + i) * dof];
    gj[i] = grad_jerk[base + (size_t)(ah + i) * dof];
  }
  float g = grad_pos[base + (size_t)(ah + 2) * dof];
  if (ah == A - 1) {
    if (use_goal) g = 0.0f;
    else g += grad_pos[base + (size_t)(ah + 3) * dof] + grad_pos[base + (size_t)(ah + 4) * dof];
  }
  float o = g;
  if (ah < A - 1) {
    o += (-0.083333333f * gv[0] + 0.666666667f * gv[1] - 0.666666667f * gv[3] + 0.083333333f * gv[4]) * i1;
    o += (-0.083333333f * ga[0] + 1.333333333f * ga[1] + (-2.5f) * ga[2] + 1.333333333f * ga[3] + (-0.083333333f) * ga[4]) * i2;
    o += (0.5f * gj[0] - 1.0f * gj[1] + 1.0f * gj[3] - 0.5f * gj[4]) * i3;
  } else if (use_goal) {
    o = 0.0f;  // the forward pass replaces the last action by the goal (reference differentiation_position_kernel.cuh:352-361)
  } else {
    o += (-0.083333333f * gv[0] + 0.583333334f * gv[1] + 0.583333334f * gv[2] - 0.083333333f * gv[3]) * i1;
    o += (-0.083333333f * ga[0] + 1.25f * ga[1] + (-1.25f) * ga[2] + 0.083333333f * ga[3]) * i2;
    o += (0.5f * gj[0] - 0.5f * gj[1] - 0.5f * gj[2] + 0.5f * gj[3]) * i3;
  }
  out_grad[((size_t)b * A + ah) * dof + d] = o;
}

// acceleration_loop_idx(_rk2)_kernel (legacy/integration_acceleration_kernel.cuh:8-135): one lane
// per (trajectory, dof) runs the recursion; the reference stages the horizon in register arrays
// sized by a template parameter (one NVRTC compile per horizon), here it streams with any horizon
__global__ void __launch_bounds__(256) integration_acceleration_kernel(
    float *out_pos, float *out_vel, float *out_acc, float *out_jerk, const float *u_acc, const float *start_pos,
    const float *start_vel, const float *start_acc, const int32_t *start_idx, const float *traj_dt, int batch,
    int horizon, int dof) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = (int)(tid / dof);
  if (b >= batch) return;
  const int d = (int)(tid - (long)b * dof);
  const int bo = start_idx[b];
  float pos = start_pos[bo * dof + d], vel = start_vel[bo * dof + d], acc = start_acc[bo * dof + d];
  size_t a = (size_t)b * horizon * dof + d;
  out_pos[a] = pos; out_vel[a] = vel; out_acc[a] = acc; out_jerk[a] = 0.0f;
  float acc_n = horizon > 1 ? u_acc[a] : 0.0f;
  for (int h = 1; h < horizon; h++) {
    const float dt = traj_dt[h];
    const float acc_next = h + 1 < horizon ? u_acc[a + dof] : 0.0f;  // prefetch: the recursion is a dependent chain
    vel = vel + acc_n * dt;
    pos = pos + vel * dt;
    a += dof;
    out_acc[a] = acc_n; out_vel[a] = vel; out_pos[a] = pos; out_jerk[a] = (acc_n - acc) / dt;
    acc = acc_n;
    acc_n = acc_next;
  }
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_launch_bspline_interpolation_forward_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk,
    float *out_dt, const float *u_position, const float *start_position,
    const float *start_velocity, const float *start_acceleration, const float *start_jerk,
    const float *goal_position, const float *goal_velocity, const float *goal_acceleration,
    const float *goal_jerk, const int32_t *start_idx, const int32_t *goal_idx,
    const float *traj_dt, const uint8_t *use_implicit_goal_state, int batch_size, int horizon,
    int dof, int n_knots, int bspline_degree, curobo_hip_stream_t stream) {
  const char *what = "launch_bspline_interpolation_forward_kernel";
  CUROBO_REQUIRE(bspline_degree >= 3 && bspline_degree <= 5, "%s: bspline_degree must be 3, 4 or 5", what);
  CUROBO_REQUIRE(n_knots >= 1 && dof >= 1 && horizon >= 2, "%s: bad n_knots/dof/horizon", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  BsFwdArgs a{};
  a.out_pos = out_position; a.out_vel = out_velocity; a.out_acc = out_acceleration; a.out_jerk = out_jerk;
  a.out_dt = out_dt; a.u = u_position;
  a.start[0] = start_position; a.start[1] = start_velocity; a.start[2] = start_acceleration; a.start[3] = start_jerk;
  a.goal[0] = goal_position; a.goal[1] = goal_velocity; a.goal[2] = goal_acceleration; a.goal[3] = goal_jerk;
  a.start_idx = start_idx; a.goal_idx = goal_idx; a.traj_dt = traj_dt; a.use_implicit_goal = use_implicit_goal_state;
  a.batch = batch_size; a.padded_horizon = horizon; a.dof = dof; a.n_knots = n_knots;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * horizon * dof, 256);
  if (bspline_degree == 3) hipLaunchKernelGGL((bspline_forward_kernel<3>), dim3(blocks), dim3(256), 0, st, a);
  else if (bspline_degree == 4) hipLaunchKernelGGL((bspline_forward_kernel<4>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bspline_forward_kernel<5>), dim3(blocks), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_bspline_interpolation_backward_kernel(
    float *out_grad_position, const float *grad_position, const float *grad_velocity,
    const float *grad_acceleration, const float *grad_jerk, const float *traj_dt,
    const int32_t *dt_idx, const uint8_t *use_implicit_goal_state, int batch_size,
    int padded_horizon, int dof, int n_knots, int bspline_degree, int use_direct_polynomial,
    curobo_hip_stream_t stream) {
  (void)use_direct_polynomial;
  const char *what = "launch_bspline_interpolation_backward_kernel";
  CUROBO_REQUIRE(bspline_degree >= 3 && bspline_degree <= 5, "%s: bspline_degree must be 3, 4 or 5", what);
  // reference cuda_core_backend/trajectory.py:157-158
  CUROBO_REQUIRE(padded_horizon - 1 >= 5, "%s: horizon must be greater than 5", what);
  CUROBO_REQUIRE(n_knots >= 1 && dof >= 1, "%s: bad n_knots/dof", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  BsBwdArgs a{};
  a.out_grad = out_grad_position;
  a.gin[0] = grad_position; a.gin[1] = grad_velocity; a.gin[2] = grad_acceleration; a.gin[3] = grad_jerk;
  a.traj_dt = traj_dt; a.dt_idx = dt_idx; a.use_implicit_goal = use_implicit_goal_state;
  a.batch = batch_size; a.padded_horizon = padded_horizon; a.dof = dof; a.n_knots = n_knots;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * n_knots * dof, 256);
  if (bspline_degree == 3) hipLaunchKernelGGL((bspline_backward_kernel<3>), dim3(blocks), dim3(256), 0, st, a);
  else if (bspline_degree == 4) hipLaunchKernelGGL((bspline_backward_kernel<4>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bspline_backward_kernel<5>), dim3(blocks), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_bspline_interpolation_single_dt_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk, float *out_dt,
    const float *knots, const float *knot_dt, const float *start_position, const float *start_velocity,
    const float *start_acceleration, const float *start_jerk, const float *goal_position,
    const float *goal_velocity, const float *goal_acceleration, const float *goal_jerk,
    const int32_t *start_idx, const int32_t *goal_idx, const float *interpolation_dt,
    const uint8_t *use_implicit_goal_state, const int32_t *interpolation_horizon, int batch_size,
    int max_out_tsteps, int dof, int n_knots, int bspline_degree, curobo_hip_stream_t stream) {
  (void)knot_dt;  // unused by the reference kernel as well (bspline_kernel.cuh:247)
  const char *what = "launch_bspline_interpolation_single_dt_kernel";
  CUROBO_REQUIRE(bspline_degree >= 3 && bspline_degree <= 5, "%s: bspline_degree must be 3, 4 or 5", what);
  CUROBO_REQUIRE(max_out_tsteps >= 1 && dof >= 1 && n_knots >= 1, "%s: bad dimensions", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  BsFwdArgs a{};
  a.out_pos = out_position; a.out_vel = out_velocity; a.out_acc = out_acceleration; a.out_jerk = out_jerk; a.out_dt = out_dt;
  a.u = knots;
  a.start[0] = start_position; a.start[1] = start_velocity; a.start[2] = start_acceleration; a.start[3] = start_jerk;
  a.goal[0] = goal_position; a.goal[1] = goal_velocity; a.goal[2] = goal_acceleration; a.goal[3] = goal_jerk;
  a.start_idx = start_idx; a.goal_idx = goal_idx; a.traj_dt = interpolation_dt; a.use_implicit_goal = use_implicit_goal_state;
  a.batch = batch_size; a.padded_horizon = max_out_tsteps; a.dof = dof; a.n_knots = n_knots;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * max_out_tsteps * dof, 256);
  if (bspline_degree == 3)
    hipLaunchKernelGGL((bspline_single_dt_kernel<3>), dim3(blocks), dim3(256), 0, st, a, interpolation_dt, interpolation_horizon, max_out_tsteps);
  else if (bspline_degree == 4)
    hipLaunchKernelGGL((bspline_single_dt_kernel<4>), dim3(blocks), dim3(256), 0, st, a, interpolation_dt, interpolation_horizon, max_out_tsteps);
  else
    hipLaunchKernelGGL((bspline_single_dt_kernel<5>), dim3(blocks), dim3(256), 0, st, a, interpolation_dt, interpolation_horizon, max_out_tsteps);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_differentiation_position_forward_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk, float *out_dt,
    const float *u_position, const float *start_position, const float *start_velocity,
    const float *start_acceleration, const float *goal_position, const float *goal_velocity,
    const float *goal_acceleration, const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,
    const uint8_t *use_implicit_goal_state, int batch_size, int horizon, int dof, curobo_hip_stream_t stream) {
  (void)goal_velocity; (void)goal_acceleration;  // unused by the reference kernel as well
  const char *what = "launch_differentiation_position_forward_kernel";
  CUROBO_REQUIRE(horizon >= 9 && dof >= 1, "%s: horizon must be >= 9 (action horizon = horizon - 4)", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * horizon * dof, 256);
  hipLaunchKernelGGL(differentiation_position_forward_kernel, dim3(blocks), dim3(256), 0, st, out_position, out_velocity,
                     out_acceleration, out_jerk, out_dt, u_position, start_position, start_velocity, start_acceleration,
                     goal_position, start_idx, goal_idx, traj_dt, use_implicit_goal_state, batch_size, horizon, dof);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_differentiation_position_backward_kernel(
    float *out_grad_position, const float *grad_position, const float *grad_velocity, const float *grad_acceleration,
    const float *grad_jerk, const float *traj_dt, const int32_t *dt_idx, const uint8_t *use_implicit_goal_state,
    int batch_size, int horizon, int dof, curobo_hip_stream_t stream) {
  const char *what = "launch_differentiation_position_backward_kernel";
  CUROBO_REQUIRE(horizon >= 9 && dof >= 1, "%s: horizon must be >= 9 (action horizon = horizon - 4)", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * (horizon - 4) * dof, 256);
  hipLaunchKernelGGL(differentiation_position_backward_kernel, dim3(blocks), dim3(256), 0, st, out_grad_position,
                     grad_position, grad_velocity, grad_acceleration, grad_jerk, traj_dt, dt_idx, use_implicit_goal_state,
                     batch_size, horizon, dof);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_integration_acceleration_kernel(
    float *out_position, float *out_velocity, float *out_acceleration, float *out_jerk, const float *u_acc,
    const float *start_position, const float *start_velocity, const float *start_acceleration,
    const int32_t *start_idx, const float *traj_dt, int batch_size, int horizon, int dof, int use_rk2,
    curobo_hip_stream_t stream) {
  (void)use_rk2;  // both reference variants run the same recursion (integration_acceleration_kernel.cuh:8-135)
  const char *what = "launch_integration_acceleration_kernel";
  CUROBO_REQUIRE(horizon >= 1 && dof >= 1, "%s: bad dimensions", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l((long)batch_size * dof, 256);
  hipLaunchKernelGGL(integration_acceleration_kernel, dim3(blocks), dim3(256), 0, st, out_position, out_velocity,
                     out_acceleration, out_jerk, u_acc, start_position, start_velocity, start_acceleration, start_idx,
                     traj_dt, batch_size, horizon, dof);
  return check_launch(what, st);
}
