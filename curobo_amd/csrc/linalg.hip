// linalg.hip -- the Levenberg-Marquardt step of the seed-IK solver: per problem
//   A = J^T J + lambda I,   delta = -A^{-1} (J^T r),   q_out = q_in + delta,
//   pred_reduction = 0.5 * delta . (lambda delta - J^T r)
// Reference: curobo/_src/optim/util/levenberg_marquardt_step.py:146-199 (an NVIDIA Warp tile
// kernel: tile_matmul + tile_cholesky + tile_cholesky_solve, one 64-thread tile per problem).
//
// This is the one dense contraction of the hot path, so it is the one place that uses the matrix
// cores: one wavefront per problem, J^T J accumulated by v_mfma_f32_16x16x4_f32 (exact fp32; A and
// B operand of a 16x16 output tile are the SAME register because both are 4 rows of J, so every
// J element is loaded once per tile column), only the upper-triangular tiles are computed.  The
// D x D system (D <= 64) is then factorised by the same wavefront in LDS (lane i owns row i,
// left-looking Cholesky) and solved with v_readlane broadcasts; nothing but q_out / pred leaves
// the CU.
#include "common.hpp"

namespace curobo_hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LmArgs {
  float *q_out, *pred;
  const float *jac, *jtr, *lam, *q_in;
  int batch, n_res, dof;
};

__device__ __forceinline__ float lane_read(float v, int src) {  // src is wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

template <int T, int WAVES>  // T x T tiles of 16 x 16 (dof <= 16 T), WAVES problems per workgroup
__global__ void __launch_bounds__(WAVES * 64) lm_step_kernel(const LmArgs a) {
  constexpr int DP = 16 * T, LD = DP + 1;
  __shared__ float s_A[WAVES][DP * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * WAVES + wave;
  if (b >= a.batch) return;
  const int D = a.dof, R = a.n_res;
  float *A = s_A[wave];
  const float *J = a.jac + (size_t)b * R * D;

  // ---- J^T J on the matrix cores
  f32x4 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = 0; j < T; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kk = lane >> 4, col = lane & 15;
  for (int k0 = 0; k0 < R; k0 += 4) {
    float x[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int r = k0 + kk, c = 16 * t + col;
      x[t] = (r < R && c < D) ? J[(size_t)r * D + c] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int j = i; j < T; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i], x[j], acc[i][j], 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg; mirror the upper tiles into the lower triangle
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = i; j < T; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * i + kk * 4 + r, cc = 16 * j + col;
        A[row * LD + cc] = acc[i][j][r];
        A[cc * LD + row] = acc[i][j][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- Cholesky A + lambda I = L L^T, lane i owns row i (left-looking; L overwrites the lower triangle)
  const float lam = a.lam[b];
  const int i = lane;
  for (int j = 0; j < D; j++) {
    float s = 0.0f;
    if (i >= j && i < D) {
      s = A[i * LD + j] + (i == j ? lam : 0.0f);
      for (int k = 0; k < j; k++) s -= A[i * LD + k] * A[j * LD + k];
    }
    const float ljj = sqrtf(lane_read(s, j));
    if (i >= j && i < D) A[i * LD + j] = (i == j) ? ljj : s / ljj;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- L y = -g, then L^T delta = y (lane i holds entry i)
  const float g = i < D ? a.jtr[(size_t)b * D + i] : 0.0f;
  float y = -g;
  for (int j = 0; j < D; j++) {
    const float yj = lane_read(y, j) / A[j * LD + j];
    if (i == j) y = yj;
    if (i > j && i < D) y -= A[i * LD + j] * yj;
  }
  float d = y;
  for (int j = D - 1; j >= 0; j--) {
    const float dj = lane_read(d, j) / A[j * LD + j];
    if (i == j) d = dj;
    if (i < j) d -= A[j * LD + i] * dj;
  }
  if (i < D) a.q_out[(size_t)b * D + i] = a.q_in[(size_t)b * D + i] + d;
  const float red = wave_sum(i < D ? d * (lam * d - g) : 0.0f);
  if (lane == 0) a.pred[b] = 0.5f * red;
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_levenberg_marquardt_step(float *joint_position_out, float *pred_reduction,
                                                      const float *jacobian, const float *jTerror,
                                                      const float *lambda_damping, const float *joint_position_in,
                                                      int batch_size, int n_residuals, int action_dim,
                                                      curobo_hip_stream_t stream) {
  const char *what = "levenberg_marquardt_step";
  CUROBO_REQUIRE(action_dim >= 1 && action_dim <= 64, "%s: action_dim=%d out of range [1,64]", what, action_dim);
  CUROBO_REQUIRE(n_residuals >= 1, "%s: n_residuals must be >= 1", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  LmArgs a{joint_position_out, pred_reduction, jacobian, jTerror, lambda_damping, joint_position_in,
           batch_size, n_residuals, action_dim};
  hipStream_t st = (hipStream_t)stream;
  if (action_dim <= 16) hipLaunchKernelGGL((lm_step_kernel<1, 4>), dim3((unsigned)ceil_div(batch_size, 4)), dim3(256), 0, st, a);
  else if (action_dim <= 32) hipLaunchKernelGGL((lm_step_kernel<2, 4>), dim3((unsigned)ceil_div(batch_size, 4)), dim3(256), 0, st, a);
  else if (action_dim <= 48) hipLaunchKernelGGL((lm_step_kernel<3, 2>), dim3((unsigned)ceil_div(batch_size, 2)), dim3(128), 0, st, a);
  else hipLaunchKernelGGL((lm_step_kernel<4, 2>), dim3((unsigned)ceil_div(batch_size, 2)), dim3(128), 0, st, a);
  return check_launch(what, st);
}
