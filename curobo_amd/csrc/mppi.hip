// mppi.hip -- the MPPI distribution update (reference optim/particle/mppi.py:201-313 with the jit
// helpers :615-757, DIAG_A covariance): per problem
//   total_p  = sum_h gamma_h cost[p,h] / gamma_0
//   w        = softmax(-total / beta)                       over the particles of the problem
//   mean'    = (1 - s_m) mean + s_m sum_p w_p a_p
//   cov'     = (1 - s_c) cov  + s_c mean_t sum_p w_p (a_p - mean)^2 + kappa ;  scale_tril' = sqrt(cov')
//   best     = a_{argmax w}                                 (SampleMode.BEST)
// The reference runs this as ~10 torch kernels (softmax, broadcasts, reductions).  Here one
// workgroup owns one problem: totals and softmax weights stay in LDS, the weighted moments are
// accumulated with the action element index on the lanes (coalesced over [horizon, dim]) in a
// fixed particle order, so the update is reproducible.
#include "common.hpp"

namespace curobo_hip {

struct MppiArgs {
  float *new_mean, *new_cov, *new_tril, *best_traj, *weights;
  const float *costs, *gamma_seq, *actions, *mean, *cov;
  int num_problems, num_particles, cost_horizon, action_horizon, action_dim;
  float beta, step_size_mean, step_size_cov, kappa;
};

__global__ void __launch_bounds__(256) mppi_update_kernel(const MppiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P = a.num_particles, Hc = a.cost_horizon, E = a.action_horizon * a.action_dim, D = a.action_dim;
  float *s_w = smem;            // [P] totals, then weights
  float *s_red = smem + P;      // [8] block reductions
  float *s_cov = s_red + 8;     // [E] per-element second moments
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const float *cost = a.costs + (size_t)b * P * Hc;
  const float g0 = a.gamma_seq[0];
  // 1. discounted total cost of every particle
  float lmin = 3.0e38f;
  int lidx = 0x7fffffff;
  for (int p = tid; p < P; p += nt) {
    float t = 0.0f;
    for (int h = 0; h < Hc; h++) t += a.gamma_seq[h] * cost[(size_t)p * Hc + h];
    t = t / g0;
    s_w[p] = t;
    if (t < lmin) { lmin = t; lidx = p; }
  }
  // block arg-min (lowest index among equal totals): max of x = -t / beta is at min t
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(lmin, off, kWave);
    const int oi = __shfl_xor(lidx, off, kWave);
    if (ov < lmin || (ov == lmin && oi < lidx)) { lmin = ov; lidx = oi; }
  }
  __shared__ int s_idx[4];
  if (lane == 0) { s_red[wave] = lmin; s_idx[wave] = lidx; }
  __syncthreads();
  float tmin = s_red[0];
  int best = s_idx[0];
  for (int w = 1; w < nw; w++)
    if (s_red[w] < tmin || (s_red[w] == tmin && s_idx[w] < best)) { tmin = s_red[w]; best = s_idx[w]; }
  __syncthreads();
  // 2. softmax(-t / beta) = exp((tmin - t) / beta) / sum
  const float inv_beta = -1.0f / a.beta;
  const float xmax = inv_beta * tmin;
  float part = 0.0f;
  for (int p = tid; p < P; p += nt) {
    const float e = expf(inv_beta * s_w[p] - xmax);
    s_w[p] = e;
    part += e;
  }
  part = wave_sum(part);
  if (lane == 0) s_red[4 + wave] = part;
  __syncthreads();
  float denom = 0.0f;
  for (int w = 0; w < nw; w++) denom += s_red[4 + w];
  for (int p = tid; p < P; p += nt) {
    const float w = s_w[p] / denom;
    s_w[p] = w;
    if (a.weights) a.weights[(size_t)b * P + p] = w;
  }
  __syncthreads();
  // 3. weighted first / second moments, element index on the lanes
  const float *act = a.actions + (size_t)b * P * E;
  for (int e = tid; e < E; e += nt) {
    const float m = a.mean[(size_t)b * E + e];
    float s1 = 0.0f, s2 = 0.0f;
    for (int p = 0; p < P; p++) {
      const float w = s_w[p], x = act[(size_t)p * E + e];
      s1 += w * x;
      const float dx = x - m;
      s2 += w * (dx * dx);
    }
    a.new_mean[(size_t)b * E + e] = (1.0f - a.step_size_mean) * m + a.step_size_mean * s1;
    s_cov[e] = s2;
    if (a.best_traj) a.best_traj[(size_t)b * E + e] = act[(size_t)best * E + e];
  }
  __syncthreads();
  // 4. diagonal covariance: mean over the action horizon, blend, floor
  for (int d = tid; d < D; d += nt) {
    float s = 0.0f;
    for (int t = 0; t < a.action_horizon; t++) s += s_cov[t * D + d];
    const float upd = s / (float)a.action_horizon;
    const float c = (1.0f - a.step_size_cov) * a.cov[(size_t)b * D + d] + a.step_size_cov * upd + a.kappa;
    a.new_cov[(size_t)b * D + d] = c;
    a.new_tril[(size_t)b * D + d] = sqrtf(c);
  }
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_mppi_update_distribution(
    float *new_mean, float *new_cov, float *new_scale_tril, float *best_traj, float *weights, const float *costs,
    const float *gamma_seq, const float *actions, const float *mean, const float *cov, int num_problems,
    int num_particles, int cost_horizon, int action_horizon, int action_dim, float beta, float step_size_mean,
    float step_size_cov, float kappa, curobo_hip_stream_t stream) {
  const char *what = "mppi_update_distribution";
  CUROBO_REQUIRE(num_particles >= 1 && cost_horizon >= 1 && action_horizon >= 1 && action_dim >= 1, "%s: bad dimensions", what);
  CUROBO_REQUIRE(beta > 0.0f, "%s: beta must be > 0", what);
  const size_t lds = ((size_t)num_particles + 8 + (size_t)action_horizon * action_dim) * sizeof(float);
  CUROBO_REQUIRE(lds <= 64 * 1024, "%s: num_particles + action_horizon * action_dim too large for LDS (%zu bytes)", what, lds);
  if (num_problems == 0) return CUROBO_HIP_OK;
  MppiArgs a{new_mean, new_cov, new_scale_tril, best_traj, weights, costs, gamma_seq, actions, mean, cov,
             num_problems, num_particles, cost_horizon, action_horizon, action_dim,
             beta, step_size_mean, step_size_cov, kappa};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mppi_update_kernel, dim3((unsigned)num_problems), dim3(256), lds, st, a);
  return check_launch(what, st);
}
