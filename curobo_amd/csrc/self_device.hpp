// self_device.hpp -- device helpers of the self-collision kernels shared with the fused rollout
// kernels.  Reference: kernels/geometry/self_collision/self_collision_helper.cuh:61-71,226-349.
#pragma once
#include "common.hpp"

namespace curobo_hip {

__device__ __forceinline__ void argmax_merge(float &v, int &k, float ov, int ok) {
  const bool take = (ov > v) || (ov == v && ok < k);
  v = take ? ov : v;
  k = take ? ok : k;
}

// one pair evaluation (reference sphere_squared_distance_fused, self_collision_helper.cuh:61-71)
__device__ __forceinline__ float pair_penetration(float4 s1, float4 s2) {
  const float r = s1.w + s2.w;
  const float dx = s1.x - s2.x, dy = s1.y - s2.y, dz = s1.z - s2.z;
  const float d2 = dx * dx + dy * dy + dz * dz;
  const float valid = (s1.w >= 0.0f && s2.w >= 0.0f) ? 1.0f : 0.0f;
  return ((r * r) - d2) * valid;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ int row16_min(int v) {
  v = min(v, dpp_i<0xB1>(v));
  v = min(v, dpp_i<0x4E>(v));
  v = min(v, dpp_i<0x141>(v));
  v = min(v, dpp_i<0x140>(v));
  return v;
}

}  // namespace curobo_hip
