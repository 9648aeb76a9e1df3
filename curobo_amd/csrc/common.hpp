// common.hpp -- shared host/device helpers for the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "curobo_hip.h"

#define CUROBO_EXPORT extern "C" __attribute__((visibility("default")))

namespace curobo_hip {

// joint types: reference kernels/kinematics/kinematics_constants.h:10-16
enum : int { J_FIXED = -1, J_X_PRISM = 0, J_Y_PRISM = 1, J_Z_PRISM = 2, J_X_ROT = 3, J_Y_ROT = 4, J_Z_ROT = 5 };

constexpr int kWave = 64;

// ---------------------------------------------------------------- host side error plumbing
int set_error(int code, const char *fmt, ...);
int check_launch(const char *what, hipStream_t stream);
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long ceil_div_l(long a, long b) { return (a + b - 1) / b; }

#define CUROBO_REQUIRE(cond, ...)                                      \
  do {                                                                 \
    if (!(cond)) return set_error(CUROBO_HIP_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

// DPP quad broadcast: every lane receives the value of lane (4*(lane/4) + K).
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);  // quad_perm:[K,K,K,K]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

// butterfly sum over groups of WIDTH consecutive lanes (WIDTH power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// DPP helpers (gfx9 encodings): quad_perm 0x00-0xff, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_zero_fill(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// wave64 all-reduce (sum), result uniform: the canonical GCN/CDNA DPP ladder -- 6 VALU adds with
// DPP operand modifiers + one v_readlane; no LDS crossbar (ds_bpermute) round trips.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_zero_fill<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_zero_fill<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_zero_fill<0x141>(v);       // row_half_mirror: 8-lane sums
  v += dpp_zero_fill<0x140>(v);       // row_mirror: every lane holds its row-of-16 sum
  v += dpp_zero_fill<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_zero_fill<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over each group of 16 consecutive lanes (a DPP row); every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_zero_fill<0xB1>(v);
  v += dpp_zero_fill<0x4E>(v);
  v += dpp_zero_fill<0x141>(v);
  v += dpp_zero_fill<0x140>(v);
  return v;
}

struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 make_f3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// sin and cos of a joint angle.  Joint angles are bounded (a few turns at most), so the argument reduction is three
// fused multiply-adds against pi/2 split in three floats and the rest two degree-7/8 minimax polynomials on
// [-pi/4, pi/4] (the classic single-precision coefficients): ~25 instructions, max abs error 9.3e-8 and max
// relative error 1.3e-7 against double precision over |x| <= 8000 (tools-free check: 2e7 random arguments).  The
// library sincosf costs ~100 instructions on this path because it carries the Payne-Hanek reduction for huge
// arguments; it stays as the fallback for |x| > 8192.
__device__ __forceinline__ void sincos_bounded(float x, float *s, float *c) {
  if (__builtin_expect(!(fabsf(x) <= 8192.0f), 0)) {
    sincosf(x, s, c);
    return;
  }
  const float k = rintf(x * 0.636619772f);
  float r = __builtin_fmaf(k, -1.5707963705062866f, x);
  r = __builtin_fmaf(k, 4.371138828673793e-08f, r);
  r = __builtin_fmaf(k, 1.7151245100058819e-15f, r);
  const int q = (int)k;
  const float r2 = r * r;
  float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
  sp = __builtin_fmaf(sp * r2, r, r);
  float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
  cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// 16-byte store that bypasses the caches' allocate-on-write (streams written once and consumed by a later launch)
typedef float float4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_float4_streaming(float4 *p, float4 v) {
#ifdef CUROBO_HIP_NO_STREAMING_STORES
  *p = v;
#else
  float4_native n;
  n.x = v.x; n.y = v.y; n.z = v.z; n.w = v.w;
  __builtin_nontemporal_store(n, reinterpret_cast<float4_native *>(p));
#endif
}

// Flat float4 copy global -> LDS by the whole workgroup with U loads per lane in flight at once (a plain
// `for (i = tid; i < n; i += nt) dst[i] = src[i]` with a run-time trip count is one global round trip PER
// ITERATION: the loop is not unrolled, the store of an iteration waits for its load).  Optional second source
// whose xyz are added (two gradient streams into one).
template <int U>
__device__ __forceinline__ void stage_float4(float4 *dst, const float4 *__restrict__ src, const float4 *__restrict__ src_b,
                                             int n, int tid, int nt) {
  if (n <= 0) return;
  for (int base = 0; base < n; base += U * nt) {
    // lanes past the end re-copy element n-1 (same value to the same word): no predicate anywhere, so the
    // compiler has nothing to hang a branch + wait per load on
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = min(base + u * nt + tid, n - 1);
      v[u] = src[i];
      if (src_b) w[u] = src_b[i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = min(base + u * nt + tid, n - 1);
      if (src_b) { v[u].x += w[u].x; v[u].y += w[u].y; v[u].z += w[u].z; }
      dst[i] = v[u];
    }
  }
}

// 3x4 row-major rigid transform applied to a point (reference kinematics_util.cuh:38-50)
__device__ __forceinline__ float4 transform_sphere(const float *C, float4 s) {
  float4 o;
  o.x = C[0] * s.x + C[1] * s.y + C[2] * s.z + C[3];
  o.y = C[4] * s.x + C[5] * s.y + C[6] * s.z + C[7];
  o.z = C[8] * s.x + C[9] * s.y + C[10] * s.z + C[11];
  o.w = s.w;
  return o;
}

// rotation block of a row-major 3x4 -> unit quaternion (x,y,z,w), w >= 0.
// reference common/quaternion_util.cuh:50-57,110-160 (4-branch form, then normalise).
__device__ __forceinline__ float4 quat_from_transform(const float *T) {
  const float t0 = T[0], t1 = T[1], t2 = T[2];
  const float t3 = T[4], t4 = T[5], t5 = T[6];
  const float t6 = T[8], t7 = T[9], t8 = T[10];
  float x, y, z, w, n, ns;
  if (t8 < 0.0f) {
    if (t0 > t4) {
      n = 1 + t0 - t4 - t8;
      ns = 0.5f / sqrtf(n);
      x = n * ns; y = (t1 + t3) * ns; z = (t6 + t2) * ns; w = -1 * (t5 - t7) * ns;
    } else {
      n = 1 - t0 + t4 - t8;
      ns = 0.5f / sqrtf(n);
      x = (t1 + t3) * ns; y = n * ns; z = (t5 + t7) * ns; w = -1 * (t6 - t2) * ns;
    }
  } else {
    if (t0 < -1 * t4) {
      n = 1 - t0 - t4 + t8;
      ns = 0.5f / sqrtf(n);
      x = (t6 + t2) * ns; y = (t5 + t7) * ns; z = n * ns; w = -1 * (t1 - t3) * ns;
    } else {
      n = 1 + t0 + t4 + t8;
      ns = 0.5f / sqrtf(n);
      x = (t5 - t7) * ns; y = (t6 - t2) * ns; z = (t1 - t3) * ns; w = -1 * n * ns;
    }
  }
  float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
  if (w < 0.0f) inv = -inv;
  return make_float4(x * inv, y * inv, z * inv, w * inv);
}

#endif  // __HIPCC__

}  // namespace curobo_hip
