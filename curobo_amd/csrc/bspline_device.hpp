// bspline_device.hpp -- uniform B-spline device functions shared by trajectory.hip and the fused
// rollout kernel.  Reference: kernels/trajectory/bspline/*.cuh (citations at each function).
#pragma once
#include "common.hpp"

namespace curobo_hip {

static __device__ __constant__ float kB3[4][4] = {{-1.0f / 6.0f, 3.0f / 6.0f, -3.0f / 6.0f, 1.0f / 6.0f},
                                           {3.0f / 6.0f, -6.0f / 6.0f, 0.0f, 4.0f / 6.0f},
                                           {-3.0f / 6.0f, 3.0f / 6.0f, 3.0f / 6.0f, 1.0f / 6.0f},
                                           {1.0f / 6.0f, 0.0f, 0.0f, 0.0f}};
static __device__ __constant__ float kB4[5][5] = {
    {1.0f / 24.0f, -4.0f / 24.0f, 6.0f / 24.0f, -4.0f / 24.0f, 1.0f / 24.0f},
    {-4.0f / 24.0f, 12.0f / 24.0f, -6.0f / 24.0f, -12.0f / 24.0f, 11.0f / 24.0f},
    {6.0f / 24.0f, -12.0f / 24.0f, -6.0f / 24.0f, 12.0f / 24.0f, 11.0f / 24.0f},
    {-4.0f / 24.0f, 4.0f / 24.0f, 6.0f / 24.0f, 4.0f / 24.0f, 1.0f / 24.0f},
    {1.0f / 24.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
static __device__ __constant__ float kB5[6][6] = {
    {-1.0f / 120.0f, 5.0f / 120.0f, -10.0f / 120.0f, 10.0f / 120.0f, -5.0f / 120.0f, 1.0f / 120.0f},
    {5.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 20.0f / 120.0f, -50.0f / 120.0f, 26.0f / 120.0f},
    {-10.0f / 120.0f, 30.0f / 120.0f, -0.0f / 120.0f, -60.0f / 120.0f, 0.0f / 120.0f, 66.0f / 120.0f},
    {10.0f / 120.0f, -20.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 50.0f / 120.0f, 26.0f / 120.0f},
    {-5.0f / 120.0f, 5.0f / 120.0f, 10.0f / 120.0f, 10.0f / 120.0f, 5.0f / 120.0f, 1.0f / 120.0f},
    {1.0f / 120.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};

// start/goal boundary knot coefficients, bspline_boundary_constraint.cuh:52-92
static __device__ __constant__ float kFix3[4][4] = {{1.0f, 1.0f, 1.0f, 1.0f},
                                             {-1.0f, 0.0f, 1.0f, 2.0f},
                                             {1.0f / 3.0f, -1.0f / 6.0f, 1.0f / 3.0f, 11.0f / 6.0f},
                                             {0.0f, 0.0f, 0.0f, 0.0f}};
static __device__ __constant__ float kFix4[4][5] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                                             {-3.0f / 2.0f, -1.0f / 2.0f, 1.0f / 2.0f, 3.0f / 2.0f, 5.0f / 2.0f},
                                             {11.0f / 12.0f, -1.0f / 12.0f, -1.0f / 12.0f, 11.0f / 12.0f, 35.0f / 12.0f},
                                             {-3.0f / 12.0f, 1.0f / 12.0f, -1.0f / 12.0f, 3.0f / 12.0f, 25.0f / 12.0f}};
static __device__ __constant__ float kFix5[4][6] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                                             {-2.0f, -1.0f, 0.0f, 1.0f, 2.0f, 3.0f},
                                             {1.75f, 0.25f, -0.25f, 0.25f, 1.75f, 4.25f},
                                             {-0.833333f, 0.083333f, 0.0f, -0.083333f, 0.833333f, 3.75f}};

// a / b of two small integers, correctly rounded (interpolate_bspline_kernel's t_mod = h / steps, bspline_kernel.cuh:118).  The
// reference builds its kernels with --prec-div=false and this library without correctly rounded fp32 division either, so the
// run-time quotient is an approximation on both -- but where a compile-time shape of the fused launch knows both operands the
// compiler folds it exactly, and shape and generic kernel were one ulp apart for step counts that are not powers of two (found
// by tests/randomised/fuzz_fused.py under run-time shapes).  Through fp64 the quotient rounds to fp32 once, the same in every
// instantiation, and equal to the oracle's IEEE division.
__device__ __forceinline__ float exact_ratio(int a, int b) { return (float)((double)a / (double)b); }

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

struct BsBwdArgs {
  float *out_grad;
  const float *gin[4];
  const float *traj_dt;
  const int32_t *dt_idx;
  const uint8_t *use_implicit_goal;
  int batch, padded_horizon, dof, n_knots;
};

// one (b, h, d) sample of the spline and its first three derivatives -> o[4]
// (reference bspline_interpolation.cuh:95-297); returns the interpolation dt of the trajectory
// (ph = padded horizon of THIS trajectory and interpolated_dt are explicit so that the single-dt
// re-interpolation, bspline_kernel.cuh:221-270, shares the code)
template <int DEG>
__device__ __forceinline__ float bspline_sample_pre(const BsFwdArgs &a, int b, int h, int d, int ph, float interpolated_dt,
                                                    int bo, int go, bool implicit_goal, float *o);

template <int DEG>
__device__ __forceinline__ float bspline_sample_at(const BsFwdArgs &a, int b, int h, int d, int ph, float interpolated_dt,
                                                   float *o) {
  const int bo = a.start_idx[b], go = a.goal_idx[b];
  return bspline_sample_pre<DEG>(a, b, h, d, ph, interpolated_dt, bo, go, a.use_implicit_goal[go] != 0, o);
}

// the same with the trajectory's indices and goal mode already in registers (a caller that samples many points of ONE
// trajectory loads them once, ahead of its other memory traffic, instead of at the head of every sample's chain)
template <int DEG>
__device__ __forceinline__ float bspline_sample_pre(const BsFwdArgs &a, int b, int h, int d, int ph, float interpolated_dt,
                                                    int bo, int go, bool implicit_goal, float *o) {
  constexpr int SUP = DEG + 1;
  const int dof = a.dof;
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
    const int sc = src < 0 ? 0 : (src < a.n_knots ? src : a.n_knots - 1);  // (clamped, unpredicated: all SUP loads in flight)
    const float kv = a.u[((size_t)b * a.n_knots + sc) * dof + d];
    knots[i] = (src < a.n_knots && src >= 0) ? kv : 0.0f;
  }
  const bool req_start = knot_idx < SUP;
  const bool req_goal = implicit_goal ? (knot_idx > a.n_knots - 1) : (knot_idx > a.n_knots);
  float t_mod = interp > 0 ? exact_ratio(h, interp) - (float)(int)(h / interp) : 0.0f;
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
  return interpolated_dt;
}

template <int DEG>
__device__ __forceinline__ float bspline_sample(const BsFwdArgs &a, int b, int h, int d, float *o) {
  return bspline_sample_at<DEG>(a, b, h, d, a.padded_horizon, a.traj_dt[a.goal_idx[b]], o);
}

// gradient of one knot (k) of one (trajectory, dof): sum over the interpolation steps of the
// support window (reference bspline_kernel.cuh:332-380, bspline_gradient_util.cuh:141-227,
// bspline_context.cuh:133-170).  gin[c] = gradient w.r.t. position/velocity/acceleration/jerk
// (NULL = zero), element (h) of this (trajectory, dof) at gin[c][addr0 + h * hstride].
template <int DEG>
__device__ __forceinline__ float bspline_knot_grad(const float *const *gin, size_t addr0, int hstride, int k, int nk,
                                                   int ph, float traj_dt, bool use_goal) {
  constexpr int SUP = DEG + 1;
  const int horizon = ph - 1;
  const int total_knots = nk + SUP;
  const int interp = horizon / total_knots;
  const int extended_horizon = total_knots * interp;
  const float knot_dt = traj_dt * (float)interp;  // bspline_common.cuh:172 (no clamp)
  const float dt2 = knot_dt * knot_dt, dt3 = knot_dt * knot_dt * knot_dt;
  const bool implicit_goal_boundary = use_goal && k >= nk - 1;
  const bool replicate_last = !use_goal && k == nk - 1;
  float total = 0.0f;
  for (int ii = 0; ii < interp; ii++) {
    float g[4][SUP];
    const int h_off = (k + 1) * interp + ii;
#pragma unroll
    for (int i = 0; i < SUP; i++) {
      const int hh = h_off + i * interp;
      const bool ld = hh < extended_horizon && !implicit_goal_boundary;
#pragma unroll
      for (int c = 0; c < 4; c++) g[c][i] = (ld && gin[c]) ? gin[c][addr0 + (size_t)hh * hstride] : 0.0f;
    }
    if (replicate_last) {  // bspline_gradient_util.cuh:181-222
#pragma unroll
      for (int i = 1; i < SUP; i++)
#pragma unroll
        for (int x = 0; x < i; x++)
#pragma unroll
          for (int c = 0; c < 4; c++) g[c][x] += g[c][i];
      if (ii == 0) {
        const float tg = gin[0][addr0 + (size_t)horizon * hstride];
#pragma unroll
        for (int x = 0; x < SUP; x++) g[0][x] += tg;
      }
    }
    const int h_idx = (k + DEG) * interp + ii;
    const float t_mod = exact_ratio(h_idx, interp) - (float)(int)(h_idx / interp);
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
  return total;
}

}  // namespace curobo_hip
