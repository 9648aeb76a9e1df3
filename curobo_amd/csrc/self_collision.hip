// self_collision.hip -- robot self-collision: max sphere-pair penetration per point, gradient
// on the arg-max pair only (reference kernels/geometry/self_collision/self_collision_kernel.cuh
// :19-297, self_collision_helper.cuh:61-349, collision_pair.cuh:13-103).
//
// gfx950 design: one wavefront owns one point when the pair list is small (franka: 818 pairs =
// 13 per lane) -- the point's spheres sit in LDS as float4, pair indices stream as packed
// int16x2 dwords (coalesced, L2 resident), and the (value, pair-index) arg-max is a pure wave64
// butterfly: no __syncthreads and no second kernel.  Large pair lists (humanoids: 1.6e5 pairs)
// give the point to the 4 waves of a workgroup and finish with a 4-entry LDS reduction; the
// reference's two-kernel map-reduce (self_collision_max_block_kernel + _max_reduce_kernel) and
// its block_batch_max_* scratch are not needed.
// Canonical tie rule: equal maxima -> lowest index in pair_locations.
#include "common.hpp"

namespace curobo_hip {

struct SelfCollArgs {
  float *out_distance;
  float *out_gradient;
  float *pair_distance;
  uint8_t *sparse_index;
  const float *robot_spheres;
  const float *offsets;
  const float *weight;
  const int16_t *pair_locations;
  int n_points, nspheres, npairs;
  int store_pair_distance, write_grad;
};

__device__ __forceinline__ void argmax_merge(float &v, int &k, float ov, int ok) {
  const bool take = (ov > v) || (ov == v && ok < k);
  v = take ? ov : v;
  k = take ? ok : k;
}

template <int WAVES_PER_POINT>
__global__ void __launch_bounds__(256) self_collision_kernel(const SelfCollArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kPointsPerBlock = 4 / WAVES_PER_POINT;
  const int S = a.nspheres, P = a.npairs;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int local_pt = wave / WAVES_PER_POINT;
  const int sub = wave % WAVES_PER_POINT;  // which slice of the pair list this wave scans
  const int n = blockIdx.x * kPointsPerBlock + local_pt;
  float4 *sph = reinterpret_cast<float4 *>(smem) + (size_t)local_pt * S;
  __shared__ float s_red_v[4];
  __shared__ int s_red_k[4];
  const bool valid_pt = n < a.n_points;

  // ---- spheres (+ padding) -> LDS; zero the rows flagged by the previous call
  //      (reference load_spheres_and_zero_gradients, self_collision_helper.cuh:151-192)
  if (valid_pt) {
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
    for (int s = sub * kWave + lane; s < S; s += kWave * WAVES_PER_POINT) {
      float4 v = src[s];
      v.w += a.offsets[s];
      sph[s] = v;
      if (a.sparse_index[(size_t)n * S + s]) {
        reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.sparse_index[(size_t)n * S + s] = 0;
      }
    }
  }
  if (WAVES_PER_POINT > 1) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

  // ---- scan this wave's slice of the pair list
  float best = 0.0f;
  int best_k = -1;
  if (valid_pt) {
    const uint32_t *pairs = reinterpret_cast<const uint32_t *>(a.pair_locations);
    for (int k = sub * kWave + lane; k < P; k += kWave * WAVES_PER_POINT) {
      const uint32_t ij = pairs[k];
      const int i = (int)(int16_t)(ij & 0xffffu), j = (int)(int16_t)(ij >> 16);
      const float4 s1 = sph[i], s2 = sph[j];
      // reference sphere_squared_distance_fused, self_collision_helper.cuh:61-71
      const float r = s1.w + s2.w;
      const float dx = s1.x - s2.x, dy = s1.y - s2.y, dz = s1.z - s2.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      const float valid = (s1.w >= 0.0f && s2.w >= 0.0f) ? 1.0f : 0.0f;
      const float f = ((r * r) - d2) * valid;
      if (a.store_pair_distance) a.pair_distance[(size_t)n * P + k] = f;
      if (f > best) { best = f; best_k = k; }
    }
  }
  // wave64 butterfly arg-max
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, kWave);
    const int ok = __shfl_xor(best_k, off, kWave);
    argmax_merge(best, best_k, ov, ok);
  }
  if (WAVES_PER_POINT > 1) {
    if (lane == 0) { s_red_v[wave] = best; s_red_k[wave] = best_k; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < WAVES_PER_POINT; w++) argmax_merge(best, best_k, s_red_v[w], s_red_k[w]);
  }
  if (!valid_pt || lane != 0) return;

  // ---- finalize (reference finalize_collision_results, self_collision_helper.cuh:277-349)
  if (best_k < 0 || best <= 0.0f) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float w = a.weight[0];
  a.out_distance[n] = 0.5f * w * best;
  if (a.write_grad) {
    const int i = a.pair_locations[2 * best_k], j = a.pair_locations[2 * best_k + 1];
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = w * (s2.x - s1.x), vy = w * (s2.y - s1.y), vz = w * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, w * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, w * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_self_collision_distance(
    float *out_distance, float *out_vec, float *pair_distance, uint8_t *sparse_index,
    const float *robot_spheres, const float *sphere_padding, const float *weight,
    const int16_t *pair_locations, float *block_batch_max_value, int16_t *block_batch_max_index,
    int num_blocks_per_batch, int max_threads_per_block, int batch_size, int horizon, int nspheres,
    int num_collision_pairs, int store_pair_distance, int compute_grad,
    curobo_hip_stream_t stream) {
  (void)block_batch_max_value; (void)block_batch_max_index; (void)num_blocks_per_batch;
  (void)max_threads_per_block;
  const char *what = "self_collision_distance";
  CUROBO_REQUIRE(nspheres >= 1 && nspheres <= 2048, "%s: nspheres=%d out of range [1,2048]", what, nspheres);
  CUROBO_REQUIRE(num_collision_pairs >= 0, "%s: negative num_collision_pairs", what);
  CUROBO_REQUIRE(((uintptr_t)pair_locations & 3) == 0, "%s: pair_locations must be 4-byte aligned", what);
  CUROBO_REQUIRE(!store_pair_distance || pair_distance, "%s: store_pair_distance needs pair_distance", what);
  const long n_points = (long)batch_size * horizon;
  if (n_points == 0) return CUROBO_HIP_OK;
  SelfCollArgs a{};
  a.out_distance = out_distance; a.out_gradient = out_vec; a.pair_distance = pair_distance;
  a.sparse_index = sparse_index; a.robot_spheres = robot_spheres; a.offsets = sphere_padding;
  a.weight = weight; a.pair_locations = pair_locations;
  a.n_points = (int)n_points; a.nspheres = nspheres; a.npairs = num_collision_pairs;
  a.store_pair_distance = store_pair_distance; a.write_grad = compute_grad;
  hipStream_t st = (hipStream_t)stream;
  if (num_collision_pairs > 4096) {
    const size_t lds = (size_t)nspheres * 16;
    hipLaunchKernelGGL((self_collision_kernel<4>), dim3((unsigned)n_points), dim3(256), lds, st, a);
  } else {
    const size_t lds = (size_t)nspheres * 16 * 4;
    hipLaunchKernelGGL((self_collision_kernel<1>), dim3((unsigned)ceil_div_l(n_points, 4)), dim3(256), lds, st, a);
  }
  return check_launch(what, st);
}
