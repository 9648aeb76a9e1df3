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
#include "common.hpp"

namespace curobo_hip {

__device__ __constant__ float kB3[4][4] = {{-1.0f / 6.0f, 3.0f / 6.0f, -3.0f / 6.0f, 1.0f / 6.0f},
                                           {3.0f / 6.0f, -6.0f / 6.0f, 0.0f, 4.0f / 6.0f},
                                           {-3.0f / 6.0f, 3.0f / 6.0f, 3.0f / 6.0f, 1.0f / 6.0f},
                                           {1.0f / 6.0f, 0.0f, 0.0f, 0.0f}};
__device__ __constant__ float kB4[5][5] = {
    {1.0f / 24.0f, -4.0f / 24.0f, 6.0f / 24.0f, -4.0f / 24.0f, 1.0f / 24.0f},
    {-4.0f / 24.0f, 12.0f / 24.0f, -6.0f / 24.0f, -12.0f / 24.0f, 11.0f / 24.0f},
    {6.0f / 24.0f, -12.0f / 24.0f, -6.0f / 24.0f, 12.0f / 24.0f, 11.0f / 24.0f},
    {-4.0f / 24.0f, 4.0f / 24.0f, 6.0f / 24.0f, 4.0f / 24.0f, 1.0f / 24.0f},
    {1.0f / 24.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
__device__ __constant__ float kB5[6][6] = {
    {-1.0f / 120.0f, 5.0f / 120.0f, -10.0f / 120.0f, 10.0f / 120.0f, -5.0f / 120.0f, 1.0f / 120.0f},
    {5.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 20.0f / 120.0f, -50.0f / 120.0f, 26.0f / 120.0f},
    {-10.0f / 120.0f, 30.0f / 120.0f, -0.0f / 120.0f, -60.0f / 120.0f, 0.0f / 120.0f, 66.0f / 120.0f},
    {10.0f / 120.0f, -20.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 50.0f / 120.0f, 26.0f / 120.0f},
    {-5.0f / 120.0f, 5.0f / 120.0f, 10.0f / 120.0f, 10.0f / 120.0f, 5.0f / 120.0f, 1.0f / 120.0f},
    {1.0f / 120.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};

// start/goal boundary knot coefficients, bspline_boundary_constraint.cuh:52-92
__device__ __constant__ float kFix3[4][4] = {{1.0f, 1.0f, 1.0f, 1.0f},
                                             {-1.0f, 0.0f, 1.0f, 2.0f},
                                             {1.0f / 3.0f, -1.0f / 6.0f, 1.0f / 3.0f, 11.0f / 6.0f},
                                             {0.0f, 0.0f, 0.0f, 0.0f}};
__device__ __constant__ float kFix4[4][5] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                                             {-3.0f / 2.0f, -1.0f / 2.0f, 1.0f / 2.0f, 3.0f / 2.0f, 5.0f / 2.0f},
                                             {11.0f / 12.0f, -1.0f / 12.0f, -1.0f / 12.0f, 11.0f / 12.0f, 35.0f / 12.0f},
                                             {-3.0f / 12.0f, 1.0f / 12.0f, -1.0f / 12.0f, 3.0f / 12.0f, 25.0f / 12.0f}};
__device__ __constant__ float kFix5[4][6] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                                             {-2.0f, -1.0f, 0.0f, 1.0f, 2.0f, 3.0f},
                                             {1.75f, 0.25f, -0.25f, 0.25f, 1.75f, 4.25f},
                                             {-0.833333f, 0.083333f, 0.0f, -0.083333f, 0.833333f, 3.75f}};

template <int DEG>
__device__ __forceinline__ float bcoef(int i, int j) {
  if (DEG == 3) return kB3[i][j];
  if (DEG == 4) return kB4[i][j];
  return kB5[i][j];
}
template <int DEG>
__device__ __forceinline__ float fixcoef(int r, int c) {
  if (DEG == 3) return kFix3[r][c];
  if (DEG == 4) return kFix4[r][c];
  return kFix5[r][c];
}

// basis of derivative order DER at t: out[i] = sum_j COEF[i][j] * d^DER/dt^DER t^(DEG-j)
template <int DEG, int DER>
__device__ __forceinline__ void basis(float t, float *out) {
  constexpr int N = DEG + 1, M = N - DER;
  float tp[M];
#pragma unroll
  for (int j = 0; j < M; j++) {
    const int pw = DEG - j;
    float coef = 1.0f;
#pragma unroll
    for (int k = 0; k < DER; k++) coef *= (float)(pw - k);
    float tv = 1.0f;
#pragma unroll
    for (int k = 0; k < pw - DER; k++) tv *= t;
    tp[j] = coef * tv;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < M; j++) acc += bcoef<DEG>(i, j) * tp[j];
    out[i] = acc;
  }
}

struct BsFwdArgs {
  float *out_pos, *out_vel, *out_acc, *out_jerk, *out_dt;
  const float *u;
  const float *start[4];
  const float *goal[4];
  const int32_t *start_idx, *goal_idx;
  const float *traj_dt;
  const uint8_t *use_implicit_goal;
  int batch, padded_horizon, dof, n_knots;
};

template <int DEG>
__global__ void __launch_bounds__(256) bspline_forward_kernel(const BsFwdArgs a) {
  constexpr int SUP = DEG + 1;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dof = a.dof, ph = a.padded_horizon;
  const int b = (int)(tid / ((long)dof * ph));
  if (b >= a.batch) return;
  const int d = (int)(tid % dof);
  const int h = (int)((tid / dof) % ph);
  const int bo = a.start_idx[b], go = a.goal_idx[b];
  const float interpolated_dt = a.traj_dt[go];
  const bool implicit_goal = a.use_implicit_goal[go] != 0;
  const int horizon = ph - 1;
  const int padded_n_knots = a.n_knots + SUP;
  const int interp = horizon / padded_n_knots;
  const float knot_dt = fmaxf(interpolated_dt, 1e-6f) * (float)interp;
  float knots[SUP];
  int knot_idx = interp > 0 ? h / interp : 0;
  bool past_end = false;
  if (knot_idx >= padded_n_knots) { knot_idx = padded_n_knots - 1; past_end = true; }
  const int start_knot = knot_idx - SUP;
#pragma unroll
  for (int i = 0; i < SUP; i++) {
    const int src = start_knot + i;
    knots[i] = (src < a.n_knots && src >= 0) ? a.u[((size_t)b * a.n_knots + src) * dof + d] : 0.0f;
  }
  const bool req_start = knot_idx < SUP;
  const bool req_goal = implicit_goal ? (knot_idx > a.n_knots - 1) : (knot_idx > a.n_knots);
  float t_mod = interp > 0 ? ((float)h / (float)interp) - (float)(int)(h / interp) : 0.0f;
  if (past_end) t_mod = 1.0f;
  const float dt2 = knot_dt * knot_dt, dt3 = knot_dt * knot_dt * knot_dt;
  if (req_start || req_goal) {
    const int ci = (req_start ? bo : go) * dof + d;
    const float *const *src = req_start ? a.start : a.goal;
    const float cpos = src[0][ci], cvel = src[1][ci], cacc = src[2][ci], cjerk = src[3][ci];
    float fixed[SUP];
#pragma unroll
    for (int i = 0; i < SUP; i++)
      fixed[i] = fixcoef<DEG>(0, i) * cpos + fixcoef<DEG>(1, i) * cvel * knot_dt +
                 fixcoef<DEG>(2, i) * cacc * dt2 + fixcoef<DEG>(3, i) * cjerk * dt3;
    // the patterns below index with run-time offsets; written as unrolled selects so that
    // knots[] / fixed[] stay in registers (no scratch)
    if (req_start) {  // assign_start_pattern: knots[i] = fixed[knot_idx + i], i < SUP - knot_idx
#pragma unroll
      for (int i = 0; i < SUP; i++)
#pragma unroll
        for (int j = 0; j < SUP; j++)
          if (j == knot_idx + i) knots[i] = fixed[j];
    } else if (implicit_goal) {  // assign_goal_pattern_implicit: knots[st + i] = fixed[i]
      const int st = SUP - (knot_idx - a.n_knots + 1);
#pragma unroll
      for (int i = 0; i < SUP; i++)
#pragma unroll
        for (int j = 0; j < SUP; j++)
          if (i == st + j) knots[i] = fixed[j];
    } else {  // assign_goal_pattern_replicate
      const int loop = knot_idx - a.n_knots;
      const int sidx = SUP - loop - 1;
      float v = knots[0];
#pragma unroll
      for (int i = 0; i < SUP; i++) v = (i == sidx) ? knots[i] : v;
#pragma unroll
      for (int i = 0; i < SUP; i++)
        if (i > sidx) knots[i] = v;
    }
  }
  float bs[SUP];
  float o[4];
  basis<DEG, 0>(t_mod, bs);
  o[0] = 0.f;
#pragma unroll
  for (int i = 0; i < SUP; i++) o[0] += knots[i] * bs[i];
  basis<DEG, 1>(t_mod, bs);
  o[1] = 0.f;
#pragma unroll
  for (int i = 0; i < SUP; i++) o[1] += knots[i] * bs[i];
  o[1] = o[1] / knot_dt;
  basis<DEG, 2>(t_mod, bs);
  o[2] = 0.f;
#pragma unroll
  for (int i = 0; i < SUP; i++) o[2] += knots[i] * bs[i];
  o[2] = o[2] / dt2;
  basis<DEG, 3>(t_mod, bs);
  o[3] = 0.f;
#pragma unroll
  for (int i = 0; i < SUP; i++) o[3] += knots[i] * bs[i];
  o[3] = o[3] / dt3;
  const size_t addr = ((size_t)b * ph + h) * dof + d;
  a.out_pos[addr] = o[0]; a.out_vel[addr] = o[1]; a.out_acc[addr] = o[2]; a.out_jerk[addr] = o[3];
  if (h == 0 && d == 0) a.out_dt[b] = interpolated_dt;
}

struct BsBwdArgs {
  float *out_grad;
  const float *gin[4];
  const float *traj_dt;
  const int32_t *dt_idx;
  const uint8_t *use_implicit_goal;
  int batch, padded_horizon, dof, n_knots;
};

template <int DEG>
__global__ void __launch_bounds__(256) bspline_backward_kernel(const BsBwdArgs a) {
  constexpr int SUP = DEG + 1;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dof = a.dof, nk = a.n_knots, ph = a.padded_horizon;
  const int b = (int)(tid / ((long)dof * nk));
  if (b >= a.batch) return;
  const int d = (int)(tid % dof);
  const int k = (int)((tid / dof) % nk);
  const int horizon = ph - 1;
  const int total_knots = nk + SUP;
  const int interp = horizon / total_knots;
  const int extended_horizon = total_knots * interp;
  const int dto = a.dt_idx[b];
  const bool use_goal = a.use_implicit_goal[dto] != 0;
  const float knot_dt = a.traj_dt[dto] * (float)interp;  // bspline_common.cuh:172 (no clamp)
  const float dt2 = knot_dt * knot_dt, dt3 = knot_dt * knot_dt * knot_dt;
  const bool implicit_goal_boundary = use_goal && k >= nk - 1;
  const bool replicate_last = !use_goal && k == nk - 1;
  const size_t addr0 = (size_t)b * ph * dof + d;
  float total = 0.0f;
  for (int ii = 0; ii < interp; ii++) {
    float g[4][SUP];
    const int h_off = (k + 1) * interp + ii;
#pragma unroll
    for (int i = 0; i < SUP; i++) {
      const int hh = h_off + i * interp;
      const bool ld = hh < extended_horizon && !implicit_goal_boundary;
#pragma unroll
      for (int c = 0; c < 4; c++) g[c][i] = ld ? a.gin[c][addr0 + (size_t)hh * dof] : 0.0f;
    }
    if (replicate_last) {  // bspline_gradient_util.cuh:181-222
#pragma unroll
      for (int i = 1; i < SUP; i++)
#pragma unroll
        for (int x = 0; x < i; x++)
#pragma unroll
          for (int c = 0; c < 4; c++) g[c][x] += g[c][i];
      if (ii == 0) {
        const float tg = a.gin[0][addr0 + (size_t)horizon * dof];
#pragma unroll
        for (int x = 0; x < SUP; x++) g[0][x] += tg;
      }
    }
    const int h_idx = (k + DEG) * interp + ii;
    const float t_mod = ((float)h_idx / (float)interp) - (float)(int)(h_idx / interp);
    float bs[SUP];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    basis<DEG, 0>(t_mod, bs);
#pragma unroll
    for (int i = 0; i < SUP; i++) s0 += g[0][i] * bs[SUP - 1 - i];
    basis<DEG, 1>(t_mod, bs);
#pragma unroll
    for (int i = 0; i < SUP; i++) s1 += g[1][i] * bs[SUP - 1 - i];
    basis<DEG, 2>(t_mod, bs);
#pragma unroll
    for (int i = 0; i < SUP; i++) s2 += g[2][i] * bs[SUP - 1 - i];
    basis<DEG, 3>(t_mod, bs);
#pragma unroll
    for (int i = 0; i < SUP; i++) s3 += g[3][i] * bs[SUP - 1 - i];
    total += s0 + (s1 / knot_dt) + (s2 / dt2) + (s3 / dt3);
  }
  a.out_grad[((size_t)b * nk + k) * dof + d] = total;
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
