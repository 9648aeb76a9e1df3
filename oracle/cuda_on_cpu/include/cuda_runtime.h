// cuda_runtime.h of oracle/cuda_on_cpu: just enough of the CUDA C++ dialect for the reference's kernels
// (/root/reference/curobo/_src/curobolib/kernels/**) to compile with g++ and run on the CPU, one fiber per CUDA thread,
// one block at a time (see simt.hpp).  TEST INFRASTRUCTURE: the reference's .cuh files are included where they
// lie, at build time, in the build container only; the products go to oracle/_ref/ (git-ignored).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __inline__ inline
#define __noinline__
#define __restrict__
#define __align__(n) alignas(n)
#define __launch_bounds__(...)
#define __shared__ static  /* blocks run one at a time: a function-local static is per-block storage */
#define __constant__ static

struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x = 1, y = 1, z = 1; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

#define CUOC_VEC(T, N2, N3, N4)                                                        \
  struct N2 { T x, y; };                                                               \
  struct N3 { T x, y, z; };                                                            \
  struct alignas(sizeof(T) * 4) N4 { T x, y, z, w; };                                  \
  inline N2 make_##N2(T x, T y) { return N2{x, y}; }                                   \
  inline N3 make_##N3(T x, T y, T z) { return N3{x, y, z}; }                           \
  inline N4 make_##N4(T x, T y, T z, T w) { return N4{x, y, z, w}; }
CUOC_VEC(float, float2, float3, float4)
CUOC_VEC(int, int2, int3, int4)
CUOC_VEC(unsigned, uint2, uint3_, uint4)
CUOC_VEC(double, double2, double3, double4)
CUOC_VEC(short, short2, short3, short4)
CUOC_VEC(unsigned short, ushort2, ushort3, ushort4)
CUOC_VEC(signed char, char2, char3, char4)
CUOC_VEC(unsigned char, uchar2, uchar3, uchar4)
inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }

namespace cuoc {
extern uint3 t_threadIdx, t_blockIdx;   // of the fiber that is running
extern dim3 t_blockDim, t_gridDim;
void syncthreads();
void syncwarp(unsigned mask);
unsigned ballot(unsigned mask, int pred);
uint32_t shfl_bits(unsigned mask, uint32_t v, int src_lane_of_me);  // value of lane `src` (lane index inside the warp)
inline void warp0_published() { syncthreads(); }  // see the Makefile: kernel_lbfgs_step's rho hand-over from warp 0 to the block
float atomic_add(float *p, float v);
int atomic_add(int *p, int v);
}  // namespace cuoc
#define threadIdx (cuoc::t_threadIdx)
#define blockIdx (cuoc::t_blockIdx)
#define blockDim (cuoc::t_blockDim)
#define gridDim (cuoc::t_gridDim)
static const int warpSize = 32;

inline void __syncthreads() { cuoc::syncthreads(); }
inline void __syncwarp(unsigned mask = 0xffffffffu) { cuoc::syncwarp(mask); }
inline unsigned __ballot_sync(unsigned mask, int pred) { return cuoc::ballot(mask, pred); }
inline unsigned __activemask() { return 0xffffffffu; }
template <typename T>
inline T cuoc_shfl(unsigned mask, T v, int src_lane) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- and 64-bit shuffles only");
  uint32_t b[2] = {0, 0};
  std::memcpy(b, &v, sizeof(T));
  b[0] = cuoc::shfl_bits(mask, b[0], src_lane);
  if (sizeof(T) == 8) b[1] = cuoc::shfl_bits(mask, b[1], src_lane);  // (as the hardware does: two 32-bit exchanges)
  T r;
  std::memcpy(&r, b, sizeof(T));
  return r;
}
inline int cuoc_lane() { return (int)((threadIdx.x + threadIdx.y * blockDim.x + threadIdx.z * blockDim.x * blockDim.y) & 31u); }
template <typename T>
inline T __shfl_sync(unsigned mask, T v, int src, int width = 32) {
  const int lane = cuoc_lane(), base = lane & ~(width - 1);
  return cuoc_shfl(mask, v, base + (src & (width - 1)));
}
template <typename T>
inline T __shfl_down_sync(unsigned mask, T v, unsigned delta, int width = 32) {
  const int lane = cuoc_lane(), base = lane & ~(width - 1);
  const int src = lane + (int)delta;
  return cuoc_shfl(mask, v, src < base + width ? src : lane);  // out of the segment: own value
}
template <typename T>
inline T __shfl_up_sync(unsigned mask, T v, unsigned delta, int width = 32) {
  const int lane = cuoc_lane(), base = lane & ~(width - 1);
  const int src = lane - (int)delta;
  return cuoc_shfl(mask, v, src >= base ? src : lane);
}
template <typename T>
inline T __shfl_xor_sync(unsigned mask, T v, int lane_mask, int width = 32) {
  const int lane = cuoc_lane(), base = lane & ~(width - 1);
  const int src = lane ^ lane_mask;
  return cuoc_shfl(mask, v, src < base + width && src >= base ? src : lane);
}
inline float atomicAdd(float *p, float v) { return cuoc::atomic_add(p, v); }
inline int atomicAdd(int *p, int v) { return cuoc::atomic_add(p, v); }

inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __fmaf_rn(float a, float b, float c) { return std::fma(a, b, c); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsqrt_rn(float a) { return std::sqrt(a); }
inline float __sinf(float a) { return std::sin(a); }
inline float __cosf(float a) { return std::cos(a); }
inline void __sincosf(float a, float *s, float *c) { *s = std::sin(a); *c = std::cos(a); }
inline void sincosf_(float a, float *s, float *c) { *s = std::sin(a); *c = std::cos(a); }
template <typename T> inline T __ldg(const T *p) { return *p; }
namespace cuoc { inline unsigned shl1(int n) { return n >= 32 || n < 0 ? 0u : 1u << n; } }  // 1 << n with the GPU's clamped shift
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
