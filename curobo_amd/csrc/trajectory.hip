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
