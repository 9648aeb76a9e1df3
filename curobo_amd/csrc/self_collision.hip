// self_collision.hip -- robot self-collision: max sphere-pair penetration per point, gradient
// on the arg-max pair only (reference kernels/geometry/self_collision/self_collision_kernel.cuh
// :19-297, self_collision_helper.cuh:61-349, collision_pair.cuh:13-103).
//
// gfx950 design: for arms (franka: 818 pairs) a point is owned by a 16-lane DPP row, 4 points per
// wave64 and 16 per workgroup: the 16 points' spheres are one contiguous float4 run (coalesced),
// the pair list (packed int16x2 dwords) is staged in LDS once per workgroup, each lane scans 4
// pairs per iteration (the list is (i,j)-sorted: sph[i] is a broadcast, sph[j] a conflict-free
// run), and the (value, pair-index) arg-max is 8 DPP row operations -- no __syncthreads after
// the staging barrier, no second kernel.  Humanoids (1.6e5 pairs, 674 spheres) give a point to
// a whole wave and stream the pair list through LDS in 4096-pair tiles shared by the 8 points
// of a workgroup; the reference's two-kernel map-reduce (self_collision_max_block_kernel +
// _max_reduce_kernel) and its block_batch_max_* scratch are not needed.
// Canonical tie rule: equal maxima -> lowest index in pair_locations.
#include "self_device.hpp"

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

// Scan pairs [k_begin, k_end) of the LDS-resident pair tile for the wave's point.  Four pairs per
// lane are in flight per iteration (pair dwords first, then the 8 sphere reads) so the two
// dependent LDS round trips overlap.  Consecutive lanes take consecutive pairs: the list is
// (i, j)-sorted, so sph[i] is mostly a broadcast and sph[j] a conflict-free run.
template <bool STORE>
__device__ __forceinline__ void scan_pairs(const uint32_t *__restrict__ s_pairs, const float4 *__restrict__ sph,
                                           int tile_base, int tile_count, int lane, float *pair_out,
                                           float &best, int &best_k) {
  constexpr int U = 4;
  for (int k0 = lane; k0 < tile_count; k0 += kWave * U) {
    uint32_t ij[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + u * kWave;
      ij[u] = s_pairs[k < tile_count ? k : 0];
    }
    float f[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = (int)(int16_t)(ij[u] & 0xffffu), j = (int)(int16_t)(ij[u] >> 16);
      f[u] = pair_penetration(sph[i], sph[j]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + u * kWave;
      if (k < tile_count) {
        if (STORE) pair_out[tile_base + k] = f[u];
        if (f[u] > best) { best = f[u]; best_k = tile_base + k; }
      }
    }
  }
}

// NWAVES waves per workgroup, one point per wave at a time, PPW points per wave in sequence.
// The pair list is staged through LDS in tiles of `tile_pairs` dwords shared by all waves
// (small robots: a single tile loaded once per workgroup; humanoids: 4096-pair tiles).
template <int NWAVES, bool STORE>
__global__ void __launch_bounds__(NWAVES * 64) self_collision_kernel(const SelfCollArgs a, int ppw, int tile_pairs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.nspheres, P = a.npairs;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  float4 *sph = reinterpret_cast<float4 *>(smem) + (size_t)wave * S;
  uint32_t *s_pairs = reinterpret_cast<uint32_t *>(reinterpret_cast<float4 *>(smem) + (size_t)NWAVES * S);
  const uint32_t *g_pairs = reinterpret_cast<const uint32_t *>(a.pair_locations);
  const bool single_tile = P <= tile_pairs;
  if (single_tile) {
    for (int k = threadIdx.x; k < P; k += NWAVES * kWave) s_pairs[k] = g_pairs[k];
    __syncthreads();
  }
  const int pt_base = (blockIdx.x * NWAVES + wave) * ppw;
  for (int it = 0; it < ppw; it++) {
    const int n = pt_base + it;
    const bool valid_pt = n < a.n_points;
    // ---- spheres (+ padding) -> this wave's LDS slot; zero rows flagged by the previous call
    //      (reference load_spheres_and_zero_gradients, self_collision_helper.cuh:151-192)
    if (valid_pt) {
      const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
      for (int s = lane; s < S; s += kWave) {
        float4 v = src[s];
        v.w += a.offsets[s];
        sph[s] = v;
        if (a.sparse_index[(size_t)n * S + s]) {
          reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
          a.sparse_index[(size_t)n * S + s] = 0;
        }
      }
    }
    float best = 0.0f;
    int best_k = -1;
    float *pair_out = STORE ? a.pair_distance + (size_t)(valid_pt ? n : 0) * P : nullptr;
    if (single_tile) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (valid_pt) scan_pairs<STORE>(s_pairs, sph, 0, P, lane, pair_out, best, best_k);
    } else {
      for (int t0 = 0; t0 < P; t0 += tile_pairs) {
        const int cnt = min(tile_pairs, P - t0);
        __syncthreads();  // previous tile fully consumed (and sphere slots written)
        for (int k = threadIdx.x; k < cnt; k += NWAVES * kWave) s_pairs[k] = g_pairs[t0 + k];
        __syncthreads();
        if (valid_pt) scan_pairs<STORE>(s_pairs, sph, t0, cnt, lane, pair_out, best, best_k);
      }
    }
    // wave64 butterfly arg-max on (value, pair index); ties -> lowest pair index
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, kWave);
      const int ok = __shfl_xor(best_k, off, kWave);
      argmax_merge(best, best_k, ov, ok);
    }
    // ---- finalize (reference finalize_collision_results, self_collision_helper.cuh:277-349)
    if (valid_pt && lane == 0) {
      if (best_k < 0 || best <= 0.0f) {
        a.out_distance[n] = 0.0f;
      } else {
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
    }
    // the sphere slot is rewritten by the next point: all lanes must be done reading it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- small pair lists (arms): 16 lanes per point, 4 points per wave, 16 points per workgroup.
// Spheres of the 16 points are one contiguous float4 run in HBM (coalesced load), the pair list
// is staged once per workgroup, and the (value, index) arg-max never leaves a 16-lane DPP row.
template <bool STORE>
__global__ void __launch_bounds__(256) self_collision_row16_kernel(const SelfCollArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kLanes = 16, kPts = 16, U = 4;
  const int S = a.nspheres, P = a.npairs;
  const int tid = threadIdx.x, grp = tid / kLanes, lane = tid % kLanes;
  float4 *sph_all = reinterpret_cast<float4 *>(smem);                       // [16][S]
  uint32_t *s_pairs = reinterpret_cast<uint32_t *>(sph_all + (size_t)kPts * S);  // [P]
  const uint32_t *g_pairs = reinterpret_cast<const uint32_t *>(a.pair_locations);
  const int pt0 = blockIdx.x * kPts;
  const int npts = min(kPts, a.n_points - pt0);
  {  // spheres (+ padding) of the workgroup's points; zero rows flagged by the previous call
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)pt0 * S;
    const size_t flat0 = (size_t)pt0 * S;
    for (int i = tid; i < npts * S; i += 256) {
      float4 v = src[i];
      v.w += a.offsets[i % S];
      sph_all[i] = v;
      if (a.sparse_index[flat0 + i]) {
        reinterpret_cast<float4 *>(a.out_gradient)[flat0 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.sparse_index[flat0 + i] = 0;
      }
    }
    for (int k = tid; k < P; k += 256) s_pairs[k] = g_pairs[k];
  }
  __syncthreads();
  const int n = pt0 + grp;
  const bool valid_pt = grp < npts;
  const float4 *sph = sph_all + (size_t)grp * S;
  float best = 0.0f;
  int best_k = 0x7fffffff;
  if (valid_pt) {
    for (int k0 = lane; k0 < P; k0 += kLanes * U) {
      uint32_t ij[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = k0 + u * kLanes;
        ij[u] = s_pairs[k < P ? k : 0];
      }
      float f[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = (int)(int16_t)(ij[u] & 0xffffu), j = (int)(int16_t)(ij[u] >> 16);
        f[u] = pair_penetration(sph[i], sph[j]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = k0 + u * kLanes;
        if (k < P) {
          if (STORE) a.pair_distance[(size_t)n * P + k] = f[u];
          if (f[u] > best) { best = f[u]; best_k = k; }
        }
      }
    }
  }
  // arg-max inside the 16-lane row: max value, then the lowest pair index that attains it
  const float m = row16_max(best);
  const int kmin = row16_min((best == m && best > 0.0f) ? best_k : 0x7fffffff);
  if (!valid_pt || lane != 0) return;
  if (kmin == 0x7fffffff || m <= 0.0f) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float w = a.weight[0];
  a.out_distance[n] = 0.5f * w * m;
  if (a.write_grad) {
    const uint32_t ij = s_pairs[kmin];
    const int i = (int)(int16_t)(ij & 0xffffu), j = (int)(int16_t)(ij >> 16);
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = w * (s2.x - s1.x), vy = w * (s2.y - s1.y), vz = w * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, w * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, w * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}


// ------------------------------------------------------------------------------------------
// Dense pair sets (humanoids: Unitree G1 has 162 k of the 227 k possible pairs of its 674 spheres).
// A pair LIST costs two 16-byte LDS gathers + an index per pair test; here the pair set is a BITMAP
// over the (i, j) matrix and the test is register tiled: a wave owns one point, lane l holds
// spheres i = 64 * slot + l of a group of G slots in registers, sphere j is ONE broadcast LDS read
// shared by G x 64 pair tests, and bit (i, j) masks the result (a disabled pair contributes +0,
// which never beats the running maximum >= 0).  Upper triangle only: a slot group starts at the
// first 32-sphere block that can hold a j above its smallest i.  Packed fp32 arithmetic on slot
// pairs.  The arg-max is tracked per (slot, 32-block) and resolved to the lexicographically first
// (i, j) among equal maxima -- the lowest index of the (i, j)-sorted pair_locations, i.e. the same
// tie rule as the list kernels.  bitmap[jb][i] (uint32, bit jj <-> pair (i, 32 jb + jj)) is built
// once per robot by the caller (curobo_amd/backends/geometry.py, from pair_locations).
struct SelfDenseArgs {
  float *out_distance;
  float *out_gradient;
  uint8_t *sparse_index;
  const float *robot_spheres;
  const float *offsets;
  const float *weight;
  const uint32_t *bitmap;  // [2 * nslots][nslots * 64]
  int n_points, nspheres, nslots, write_grad;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// penetration of spheres (a, b) of two slots against sphere j, masked by bit jj of the slots' words
__device__ __forceinline__ v2f dense_pair2(v2f x, v2f y, v2f z, v2f r, float4 sj) {
  const v2f dx = x - sj.x, dy = y - sj.y, dz = z - sj.z, rr = r + sj.w;
  return rr * rr - (dx * dx + dy * dy + dz * dz);
}
__device__ __forceinline__ float mask_f(float v, uint32_t word, int jj) {  // v if bit jj of word else +0
  const int m = __builtin_amdgcn_sbfe((int)word, jj, 1);  // 0 or -1
  return __int_as_float(__float_as_int(v) & m);
}

template <int G>
__global__ void __launch_bounds__(256) self_collision_dense_kernel(const SelfDenseArgs a) {
  static_assert(G == 4, "slot groups of four (two packed pairs)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.nspheres, NS = a.nslots, SL = NS * 64, NB = NS * 2;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  float4 *sph = reinterpret_cast<float4 *>(smem) + (size_t)wave * SL;
  const int n = blockIdx.x * (blockDim.x / kWave) + wave;
  if (n >= a.n_points) return;  // waves are independent: only wave-level fences below
  const float qnan = __builtin_nanf("");
  {  // spheres (+ padding) -> this wave's LDS slot; disabled / padding spheres carry a NaN radius (lose every max)
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
    for (int s = lane; s < SL; s += kWave) {
      float4 v = make_float4(0.f, 0.f, 0.f, qnan);
      if (s < S) {
        v = src[s];
        v.w += a.offsets[s];
        if (!(v.w >= 0.0f)) v.w = qnan;  // pair_penetration: pairs with a negative (padded) radius contribute 0
        if (a.sparse_index[(size_t)n * S + s]) {
          reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
          a.sparse_index[(size_t)n * S + s] = 0;
        }
      }
      sph[s] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float best = 0.0f;
  int best_code = -1;  // slot * NB + jb of the first (slot-major, then block) 32-block that attains `best`
  for (int g0 = 0; g0 < NS; g0 += G) {
    float4 own[G];
#pragma unroll
    for (int t = 0; t < G; t++) own[t] = g0 + t < NS ? sph[(g0 + t) * 64 + lane] : make_float4(0.f, 0.f, 0.f, qnan);
    const v2f x01 = {own[0].x, own[1].x}, y01 = {own[0].y, own[1].y}, z01 = {own[0].z, own[1].z}, r01 = {own[0].w, own[1].w};
    const v2f x23 = {own[2].x, own[3].x}, y23 = {own[2].y, own[3].y}, z23 = {own[2].z, own[3].z}, r23 = {own[2].w, own[3].w};
    float gbest[G] = {0.f, 0.f, 0.f, 0.f};
    int gjb[G] = {0, 0, 0, 0};
    for (int jb = 2 * g0; jb < NB; jb++) {
      uint32_t w[G];
#pragma unroll
      for (int t = 0; t < G; t++) w[t] = g0 + t < NS ? a.bitmap[(size_t)jb * SL + (g0 + t) * 64 + lane] : 0u;
      if (__ballot((w[0] | w[1] | w[2] | w[3]) != 0u) == 0ull) continue;  // no enabled pair in this tile (uniform)
      const float4 *sj = sph + jb * 32;
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      if (__ballot((w[0] & w[1] & w[2] & w[3]) != 0xffffffffu) == 0ull) {
        // every pair of the tile is enabled (spheres of links far apart in the tree): no mask arithmetic
#pragma unroll
        for (int jj = 0; jj < 32; jj++) {
          const float4 s = sj[jj];
          const v2f p01 = dense_pair2(x01, y01, z01, r01, s), p23 = dense_pair2(x23, y23, z23, r23, s);
          b0 = fmaxf(b0, p01.x);
          b1 = fmaxf(b1, p01.y);
          b2 = fmaxf(b2, p23.x);
          b3 = fmaxf(b3, p23.y);
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < 32; jj++) {
          const float4 s = sj[jj];
          const v2f p01 = dense_pair2(x01, y01, z01, r01, s), p23 = dense_pair2(x23, y23, z23, r23, s);
          b0 = fmaxf(b0, mask_f(p01.x, w[0], jj));
          b1 = fmaxf(b1, mask_f(p01.y, w[1], jj));
          b2 = fmaxf(b2, mask_f(p23.x, w[2], jj));
          b3 = fmaxf(b3, mask_f(p23.y, w[3], jj));
        }
      }
      const float bb[G] = {b0, b1, b2, b3};
#pragma unroll
      for (int t = 0; t < G; t++)
        if (bb[t] > gbest[t]) { gbest[t] = bb[t]; gjb[t] = jb; }  // strict: the first block of a slot wins ties
    }
#pragma unroll
    for (int t = 0; t < G; t++)
      if (gbest[t] > best) { best = gbest[t]; best_code = (g0 + t) * NB + gjb[t]; }  // strict: the lowest slot wins ties
  }
  // ---- arg-max over the wave; lanes that attain it look up the first j of their winning block
  float m = best;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  int key = 0x7fffffff;
  if (m > 0.0f && best == m) {
    const int slot = best_code / NB, jb = best_code - slot * NB;
    const float4 o = sph[slot * 64 + lane];
    const uint32_t word = a.bitmap[(size_t)jb * SL + slot * 64 + lane];
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, orr = {o.w, o.w};
    float bv = 0.0f;
    int bj = -1;
    for (int jj = 0; jj < 32; jj++) {
      const float v = mask_f(dense_pair2(ox, oy, oz, orr, sph[jb * 32 + jj]).x, word, jj);
      if (v > bv) { bv = v; bj = jb * 32 + jj; }
    }
    if (bj >= 0) key = ((slot * 64 + lane) << 10) | bj;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, kWave));
  if (lane != 0) return;
  if (!(m > 0.0f) || key == 0x7fffffff) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float wgt = a.weight[0];
  a.out_distance[n] = 0.5f * wgt * m;
  if (a.write_grad) {
    const int i = key >> 10, j = key & 1023;
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = wgt * (s2.x - s1.x), vy = wgt * (s2.y - s1.y), vz = wgt * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, wgt * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, wgt * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}

// ------------------------------------------------------------------------------------------
// The same pair set with a BROAD PHASE in front.  The spheres of a robot file come link by link, so 16 consecutive
// spheres sit close together: per point, every 16-sphere block gets an axis-aligned box (centres +- radii), and a
// 16 x 16 tile of the pair matrix is only evaluated when the boxes of its two blocks overlap -- otherwise no pair of
// it can penetrate, and only positive penetrations count (result preserving).  For a humanoid in a typical pose one
// in seven of the tiles that hold enabled pairs survives (Unitree G1: 107 of 763; 32 x 32 tiles: 63 of 224 = twice
// the pair tests).  `tiles` = the (ib | jb << 8) list of tiles with at least one enabled pair (built once per robot
// next to the bitmap).  A wavefront owns a point; the surviving tiles are compacted and taken four at a time, one per
// 16-lane row: lane (row, li) keeps sphere i = 16 ib + li in registers and streams the 16 spheres of block jb from LDS.
constexpr int kTile = 16;
struct SelfTilesArgs {
  SelfDenseArgs d;
  const int32_t *tiles;
  int n_tiles;
  const uint8_t *lane_masks;  // optional [n_tiles][64]: the tile's listed pairs in the matrix-core result layout (see the mfma kernel)
};

__device__ __forceinline__ float pair_pen(float4 o, float4 sj) {
  const float dx = o.x - sj.x, dy = o.y - sj.y, dz = o.z - sj.z, rr = o.w + sj.w;
  return rr * rr - (dx * dx + dy * dy + dz * dz);
}

__global__ void __launch_bounds__(256) self_collision_tiles_kernel(const SelfTilesArgs t) {
  const SelfDenseArgs &a = t.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.nspheres, NS = a.nslots, SL = NS * 64, NB = SL / kTile;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const size_t wave_floats = (size_t)SL * 4 + (size_t)NB * 8 + (size_t)((t.n_tiles + 3) & ~3);
  float4 *sph = reinterpret_cast<float4 *>(smem + wave * wave_floats);
  float *box = reinterpret_cast<float *>(sph + SL);       // [NB][8]: lo xyz, -, hi xyz, -
  int *kept = reinterpret_cast<int *>(box + NB * 8);      // surviving tiles
  const int n = blockIdx.x * (blockDim.x / kWave) + wave;
  if (n >= a.n_points) return;  // waves are independent: only wave-level fences below
  const float qnan = __builtin_nanf("");
  const int row = lane >> 4, li = lane & 15;
  {  // spheres (+ padding) -> LDS, stale gradient rows cleared (as the other kernels), block boxes on the way
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
    for (int sl = 0; sl < NS; sl++) {
      const int s = sl * 64 + lane;
      float4 v = make_float4(0.f, 0.f, 0.f, qnan);
      if (s < S) {
        v = src[s];
        v.w += a.offsets[s];
        if (!(v.w >= 0.0f)) v.w = qnan;
        if (a.sparse_index[(size_t)n * S + s]) {
          reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
          a.sparse_index[(size_t)n * S + s] = 0;
        }
      }
      sph[s] = v;
      const bool on = v.w == v.w;  // disabled / padding spheres: an empty box
      const float big = 3.0e38f;
      const float lx = -row16_max(on ? v.w - v.x : -big), ly = -row16_max(on ? v.w - v.y : -big), lz = -row16_max(on ? v.w - v.z : -big);
      const float hx = row16_max(on ? v.x + v.w : -big), hy = row16_max(on ? v.y + v.w : -big), hz = row16_max(on ? v.z + v.w : -big);
      if (li == 0) {
        float *b = box + (sl * 4 + row) * 8;
        b[0] = lx; b[1] = ly; b[2] = lz; b[4] = hx; b[5] = hy; b[6] = hz;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- broad phase: tiles whose block boxes overlap, compacted in list order
  int count = 0;
  for (int c0 = 0; c0 < t.n_tiles; c0 += kWave) {
    const int c = c0 + lane;
    bool keep = false;
    int tl = 0;
    if (c < t.n_tiles) {
      tl = t.tiles[c];
      const float *bi = box + (tl & 0xff) * 8, *bj = box + (tl >> 8) * 8;
      keep = bi[0] <= bj[4] && bj[0] <= bi[4] && bi[1] <= bj[5] && bj[1] <= bi[5] && bi[2] <= bj[6] && bj[2] <= bi[6];
    }
    const unsigned long long m = __ballot(keep);
    if (keep) kept[count + __popcll(m & ((1ull << lane) - 1ull))] = tl;
    count += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- narrow phase: four tiles per round, one per 16-lane row
  float bv = 0.0f;
  int bkey = 0x7fffffff;
  for (int t0 = 0; t0 < count; t0 += 4) {
    const bool has = t0 + row < count;
    const int tl = kept[has ? t0 + row : t0];
    const int ib = tl & 0xff, jb = tl >> 8;
    const int i = ib * kTile + li;
    const float4 own = sph[i];
    // bit jj of bitmap[j / 32][i] <-> pair (i, j): the 16 bits of this tile's j block
    const uint32_t word = has ? (a.bitmap[(size_t)(jb >> 1) * SL + i] >> ((jb & 1) * 16)) & 0xffffu : 0u;
    if (__ballot(word != 0u) == 0ull) continue;
    const float4 *sj = sph + jb * kTile;
    float tm = 0.0f;
#pragma unroll
    for (int jj = 0; jj < kTile; jj++) tm = fmaxf(tm, mask_f(pair_pen(own, sj[jj]), word, jj));
    if (tm > 0.0f && tm >= bv) {  // the lane's best so far, or a tie with it: the first j of this tile that attains the
      // maximum.  Value and index come from THIS loop (the unrolled loop above may contract the products differently,
      // so an equality test against its maximum can fail by an ulp)
      float tv = 0.0f;
      int jf = 0;
#pragma unroll 1
      for (int jj = 0; jj < kTile; jj++) {
        const float v = mask_f(pair_pen(own, sj[jj]), word, jj);
        if (v > tv) { tv = v; jf = jj; }
      }
      if (tv > 0.0f && tv >= bv) {
        const int key = (i << 10) | (jb * kTile + jf);
        if (tv > bv || key < bkey) bkey = key;
        bv = tv;
      }
    }
  }
  // ---- arg-max over the wave: largest penetration, then the lexicographically first (i, j)
  float m = bv;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  int key = (m > 0.0f && bv == m) ? bkey : 0x7fffffff;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, kWave));
  if (lane != 0) return;
  if (!(m > 0.0f) || key == 0x7fffffff) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float wgt = a.weight[0];
  a.out_distance[n] = 0.5f * wgt * m;
  if (a.write_grad) {
    const int i = key >> 10, j = key & 1023;
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = wgt * (s2.x - s1.x), vy = wgt * (s2.y - s1.y), vz = wgt * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, wgt * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, wgt * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}

// ------------------------------------------------------------------------------------------
// TWO-LEVEL broad phase.  Measured on the Unitree G1 (tools: broad-phase statistics in DESIGN.md): of the 763 tiles of
// 16 x 16 pairs that hold listed pairs ~100 survive the block-box test -- 25.7 k pair tests per point -- but only ~29 listed
// pairs really penetrate.  Boxes of FOUR consecutive spheres are four times tighter: testing the 16 sub-tiles of every
// surviving tile against them leaves ~260 sub-tiles of 4 x 4 = 4.2 k pair tests per point.  The sub-tile boxes cost
// nothing extra (the second step of the 16-lane DPP maximum is the 4-lane maximum).  Narrow phase: the surviving sub-tiles
// are compacted, four per round, lane (t, pi, pj) tests ONE pair (i = 4 ib4 + pi, j = 4 jb4 + pj) with its bit of the pair
// bitmap; arg-max with the list's tie rule, (i, j) lexicographic.  Same pair_pen as the one-level kernel: same values.
constexpr int kT2Waves = 4;   // wavefronts that share one point (its spheres and boxes in LDS)
#ifndef CUROBO_T2_RING
#define CUROBO_T2_RING 72
#endif
constexpr int kT2Ring = CUROBO_T2_RING;  // sub-tiles a wavefront may hold before its narrow phase runs: < 4 left over + <= 64 of a trip.  (A 512-entry
                                         // ring drained after four trips was 25 % slower: 8 KB of LDS per point are two points per CU)
constexpr int kT2Box = 6;    // floats per box: lo xyz, hi xyz (every byte of LDS per point is occupancy: eight points per CU at 20 KB)
__host__ __device__ inline size_t tiles2_kept_cap(int n_tiles) { return (size_t)((n_tiles + kT2Waves * 64 - 1) / (kT2Waves * 64)) * 64; }
__host__ __device__ inline size_t tiles2_lds_floats(int nslots, int n_tiles) {
  const size_t SL = (size_t)nslots * 64;
  return SL * 4 + (SL / kTile) * kT2Box + (SL / 4) * kT2Box + kT2Waves * (tiles2_kept_cap(n_tiles) + kT2Ring) + 2 * kT2Waves;
}

// One POINT per workgroup of four wavefronts.  A wavefront per point is latency bound (a dependent chain of LDS and
// L2 round trips) and a CU only holds as many points as their spheres fit in its LDS (G1: 6-9): the SIMDs idle.  Four
// wavefronts that share the point's staged spheres and boxes split the staging, the tile list and the narrow phase, so a CU
// holds the same number of points with four times the wavefronts, and a point takes a quarter of the time.
__global__ void __launch_bounds__(kT2Waves * 64) self_collision_tiles2_kernel(const SelfTilesArgs t) {
  const SelfDenseArgs &a = t.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.nspheres, NS = a.nslots, SL = NS * 64, NB = SL / kTile, NB4 = SL / 4;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int kept_cap = (int)tiles2_kept_cap(t.n_tiles);
  float4 *sph = reinterpret_cast<float4 *>(smem);
  float *box = reinterpret_cast<float *>(sph + SL);       // [NB][6]: lo xyz, hi xyz
  float *box4 = box + NB * kT2Box;                        // [NB4][6]
  int *kept = reinterpret_cast<int *>(box4 + NB4 * kT2Box) + wave * (kept_cap + kT2Ring);  // this wavefront's surviving 16 x 16 tiles
  int *ring = kept + kept_cap;                            // ... and its surviving 4 x 4 sub-tiles
  float *red = box4 + NB4 * kT2Box + kT2Waves * (kept_cap + kT2Ring);  // [kT2Waves] (penetration, key) per wavefront
  const int n = blockIdx.x;
  const float qnan = __builtin_nanf("");
  const int row = lane >> 4, li = lane & 15;
  // this wavefront's tiles of the level-1 list: requested now, used after the staging (the list is 3 KB: L2)
  constexpr int kMaxTileTrips = 8;  // <= 8 * 256 = 2048 listed tiles (64 x 64 blocks of 16 spheres hold 2080)
  int my_tiles[kMaxTileTrips];
#pragma unroll
  for (int u = 0; u < kMaxTileTrips; u++) {
    const int c = (u * kT2Waves + wave) * kWave + lane;
    my_tiles[u] = c < t.n_tiles ? t.tiles[c] : -1;
  }
  {  // spheres (+ padding) -> LDS, stale gradient rows cleared, boxes of 4 and of 16 consecutive spheres on the way
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
    constexpr int kMaxSlots = 4;  // nslots <= 16, four wavefronts
    float4 sv[kMaxSlots];
    float off[kMaxSlots];
    uint8_t dirty[kMaxSlots];
#pragma unroll
    for (int u = 0; u < kMaxSlots; u++) {  // every load of the wavefront in flight before the first use
      const int s = (u * kT2Waves + wave) * 64 + lane;
      const bool in = u * kT2Waves + wave < NS && s < S;
      sv[u] = in ? src[s] : make_float4(0.f, 0.f, 0.f, qnan);
      off[u] = in ? a.offsets[s] : 0.0f;
      dirty[u] = in ? a.sparse_index[(size_t)n * S + s] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < kMaxSlots; u++) {
      const int sl = u * kT2Waves + wave;
      if (sl >= NS) break;
      const int s = sl * 64 + lane;
      float4 v = sv[u];
      if (s < S) {
        v.w += off[u];
        if (!(v.w >= 0.0f)) v.w = qnan;
        if (dirty[u]) {
          reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
          a.sparse_index[(size_t)n * S + s] = 0;
        }
      }
      sph[s] = v;
      const bool on = v.w == v.w;  // disabled / padding spheres: an empty box
      const float big = 3.0e38f;
      float e[6] = {on ? v.w - v.x : -big, on ? v.w - v.y : -big, on ? v.w - v.z : -big,
                    on ? v.x + v.w : -big, on ? v.y + v.w : -big, on ? v.z + v.w : -big};
#pragma unroll
      for (int c = 0; c < 6; c++) {  // quad maximum, then the row's
        e[c] = fmaxf(e[c], dpp_f<0xB1>(e[c]));
        e[c] = fmaxf(e[c], dpp_f<0x4E>(e[c]));
      }
      if ((lane & 3) == 0) {
        float *b = box4 + (size_t)(s >> 2) * kT2Box;
        b[0] = -e[0]; b[1] = -e[1]; b[2] = -e[2]; b[3] = e[3]; b[4] = e[4]; b[5] = e[5];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        e[c] = fmaxf(e[c], dpp_f<0x141>(e[c]));
        e[c] = fmaxf(e[c], dpp_f<0x140>(e[c]));
      }
      if (li == 0) {
        float *b = box + (sl * 4 + row) * kT2Box;
        b[0] = -e[0]; b[1] = -e[1]; b[2] = -e[2]; b[3] = e[3]; b[4] = e[4]; b[5] = e[5];
      }
    }
  }
  __syncthreads();
  // ---- level 1: this wavefront's share of the tiles whose block boxes overlap, compacted in list order
  int count = 0;
#pragma unroll
  for (int u = 0; u < kMaxTileTrips; u++) {
    if ((u * kT2Waves + wave) * kWave >= t.n_tiles) break;
    const int tl = my_tiles[u];
    bool keep = false;
    if (tl >= 0) {
      const float *bi = box + (tl & 0xff) * kT2Box, *bj = box + (tl >> 8) * kT2Box;
      keep = bi[0] <= bj[3] && bj[0] <= bi[3] && bi[1] <= bj[4] && bj[1] <= bi[4] && bi[2] <= bj[5] && bj[2] <= bi[5];
    }
    const unsigned long long m = __ballot(keep);
    if (keep) kept[count + __popcll(m & ((1ull << lane) - 1ull))] = tl;
    count += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- level 2: four tiles = 64 sub-tiles per round, one per lane.  A surviving sub-tile fetches its 16 bits of the pair
  // bitmap (one 16-byte load: the words of its four spheres i) -- sub-tiles without a listed pair drop out here and the
  // narrow phase needs no memory access besides LDS.  Ring entry: ib4 | jb4 << 8 | bits << 16.
  float bv = 0.0f;
  int bkey = 0x7fffffff;
  int n4 = 0;  // sub-tiles in the ring (uniform)
  const int sub_t = lane >> 4, pi = (lane >> 2) & 3, pj = lane & 3;
  auto narrow = [&](int first) {  // ring entries first .. first + 3 (entries beyond n4 are skipped by `has`)
    const bool has = first + sub_t < n4;
    const uint32_t e = (uint32_t)ring[has ? first + sub_t : 0];
    const int i = (int)(e & 0xffu) * 4 + pi, j = (int)((e >> 8) & 0xffu) * 4 + pj;
    const float v = mask_f(pair_pen(sph[i], sph[j]), has ? e : 0u, 16 + pi * 4 + pj);
    const int key = (i << 10) | j;
    if (v > 0.0f && (v > bv || (v == bv && key < bkey))) { bv = v; bkey = key; }
  };
  auto drain = [&](bool all) {  // the narrow phase over the ring; `all`: also the last, incomplete group
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int first = 0;
    for (; first + 16 <= n4; first += 16) {  // four independent groups per trip: their LDS reads go out together
      narrow(first); narrow(first + 4); narrow(first + 8); narrow(first + 12);
    }
    for (; first + 4 <= n4; first += 4) narrow(first);
    if (all && first < n4) { narrow(first); first = n4; }
    const int left = n4 - first;
    int keep_e = 0;
    if (left > 0 && first > 0) {  // move the < 4 leftovers to the front
      if (lane < left) keep_e = ring[first + lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < left) ring[lane] = keep_e;
    }
    n4 = left;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // One trip = four kept tiles = 64 sub-tiles, one per lane.  The next trip's box tests and bitmap loads are issued before
  // this trip's survivors go through the ring and the narrow phase: the L2 round trip of the words overlaps LDS-only work.
  struct Trip { bool keep; int ib4, jb4; uint4 w; };
  auto fetch = [&](int t0) {
    Trip tr;
    const int ti = t0 + sub_t;
    const bool has = ti < count;
    const int tl = kept[has ? ti : 0];
    tr.ib4 = (tl & 0xff) * 4 + pi;
    tr.jb4 = (tl >> 8) * 4 + pj;
    const float *bi = box4 + (size_t)tr.ib4 * kT2Box, *bj = box4 + (size_t)tr.jb4 * kT2Box;
    // (the diagonal tiles list pairs i < j only: a sub-tile strictly below the diagonal holds none of them)
    tr.keep = has && tr.ib4 <= tr.jb4 && bi[0] <= bj[3] && bj[0] <= bi[3] && bi[1] <= bj[4] && bj[1] <= bi[4] &&
              bi[2] <= bj[5] && bj[2] <= bi[5];
    // bit (jb4 * 4 + pj') & 31 of bitmap[(jb4 * 4) >> 5][ib4 * 4 + pi'] for the 4 x 4 pairs of the sub-tile
    tr.w = tr.keep ? *reinterpret_cast<const uint4 *>(a.bitmap + (size_t)(tr.jb4 >> 3) * SL + tr.ib4 * 4) : make_uint4(0u, 0u, 0u, 0u);
    return tr;
  };
  Trip nxt = fetch(0);
  for (int t0 = 0; t0 < count; t0 += 4) {
    const Trip cur = nxt;
    if (t0 + 4 < count) nxt = fetch(t0 + 4);
    const int sh = (cur.jb4 & 7) * 4;
    const uint32_t bits = ((cur.w.x >> sh) & 0xfu) | (((cur.w.y >> sh) & 0xfu) << 4) | (((cur.w.z >> sh) & 0xfu) << 8) |
                          (((cur.w.w >> sh) & 0xfu) << 12);
    const bool k2 = cur.keep && bits != 0u;
    const unsigned long long m = __ballot(k2);
    if (k2) ring[n4 + __popcll(m & ((1ull << lane) - 1ull))] = (int)((uint32_t)cur.ib4 | ((uint32_t)cur.jb4 << 8) | (bits << 16));
    n4 += __popcll(m);
    if (n4 + 64 > kT2Ring) drain(false);
  }
  drain(true);
  // ---- arg-max: largest penetration, then the lexicographically first (i, j); over the wavefront, then over the four
  float m = bv;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  int key = (m > 0.0f && bv == m) ? bkey : 0x7fffffff;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, kWave));
  if (lane == 0) {
    red[wave * 2] = m;
    reinterpret_cast<int *>(red)[wave * 2 + 1] = key;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  m = 0.0f;
  key = 0x7fffffff;
  for (int w = 0; w < kT2Waves; w++) {
    const float mw = red[w * 2];
    const int kw = reinterpret_cast<const int *>(red)[w * 2 + 1];
    if (mw > m || (mw == m && kw < key)) { m = mw; key = kw; }
  }
  if (!(m > 0.0f) || key == 0x7fffffff) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float wgt = a.weight[0];
  a.out_distance[n] = 0.5f * wgt * m;
  if (a.write_grad) {
    const int i = key >> 10, j = key & 1023;
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = wgt * (s2.x - s1.x), vy = wgt * (s2.y - s1.y), vz = wgt * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, wgt * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, wgt * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}

// ------------------------------------------------------------------------------------------
// MATRIX-CORE narrow phase.  After the level-1 block-box test a point keeps ~100 of its 763 listed 16 x 16 tiles; the
// two-level kernel above then spends most of its instructions on the second level (sub-tile boxes, ring compaction, the
// 4 x 4 pair tests) while the matrix pipe idles.  A 16 x 16 tile of penetrations is ONE v_mfma_f32_16x16x4_f32:
//   (r_i + r_j)^2 - |c_i - c_j|^2 = [2 c_i . c_j + 2 r_i r_j] + (r_i^2 - |c_i|^2) + (r_j^2 - |c_j|^2)
// K = 4 = (x, y, z, r): A = 2 (x, y, z, r)_i, B = (x, y, z, r)_j, the row / column terms as the accumulator's initial
// value.  fp32 operands and accumulation (the f32 MFMA is an fmaf chain), but the expanded form cancels (|c|^2 ~ 1 against
// penetrations ~1e-3), so its result only CULLS: a listed pair whose tile value exceeds -kMfmaSlack is evaluated again
// with pair_pen -- the same function, the same bits as the other kernels -- and only those values enter the arg-max.
// Same outputs as self_collision_tiles2_kernel, bit for bit (test_self_collision_dense_bitmap_kernel_c4_size).
// Reference: self_collision_kernel.cuh:113-297 (map-reduce max over the pair list).
constexpr int kTmWaves = 4;
constexpr float kMfmaSlack = 4.0e-5f;  // m^2: > the rounding of five fp32 terms of magnitude <= 16 (coordinates within +-2 m)
__host__ __device__ inline size_t tilesm_kept_cap(int n_tiles) { return (size_t)((n_tiles + kTmWaves * 64 - 1) / (kTmWaves * 64)) * 64; }
__host__ __device__ inline size_t tilesm_lds_floats(int nslots, int n_tiles) {
  const size_t SL = (size_t)nslots * 64;
  return SL * 4 + SL + (SL / kTile) * kT2Box + kTmWaves * tilesm_kept_cap(n_tiles) + 2 * kTmWaves;
}
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kTmWaves * 64) self_collision_tiles_mfma_kernel(const SelfTilesArgs t) {
  const SelfDenseArgs &a = t.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.nspheres, NS = a.nslots, SL = NS * 64, NB = SL / kTile;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int kept_cap = (int)tilesm_kept_cap(t.n_tiles);
  float4 *sph = reinterpret_cast<float4 *>(smem);
  float *aterm = reinterpret_cast<float *>(sph + SL);     // [SL]: h = (r^2 - |c|^2) / 2 (NaN for disabled / padding spheres)
  float *box = aterm + SL;                                // [NB][6]: lo xyz, hi xyz of 16 consecutive spheres
  int *kept = reinterpret_cast<int *>(box + NB * kT2Box) + wave * kept_cap;  // this wavefront's surviving tiles
  float *red = box + NB * kT2Box + kTmWaves * kept_cap;   // [kTmWaves] (penetration, key) per wavefront
  const int n = blockIdx.x;
  const float qnan = __builtin_nanf("");
  const int row = lane >> 4, li = lane & 15;
  constexpr int kMaxTileTrips = 8;  // <= 8 * 256 = 2048 listed tiles
  int my_tiles[kMaxTileTrips];
#pragma unroll
  for (int u = 0; u < kMaxTileTrips; u++) {
    const int c = (u * kTmWaves + wave) * kWave + lane;
    my_tiles[u] = c < t.n_tiles ? t.tiles[c] : -1;
  }
  {  // spheres (+ padding) -> LDS with their row / column term, stale gradient rows cleared, boxes of 16 consecutive spheres
    const float4 *src = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)n * S;
    constexpr int kMaxSlots = 4;  // nslots <= 16, four wavefronts
    float4 sv[kMaxSlots];
    float off[kMaxSlots];
    uint8_t dirty[kMaxSlots];
#pragma unroll
    for (int u = 0; u < kMaxSlots; u++) {
      const int s = (u * kTmWaves + wave) * 64 + lane;
      const bool in = u * kTmWaves + wave < NS && s < S;
      sv[u] = in ? src[s] : make_float4(0.f, 0.f, 0.f, qnan);
      off[u] = in ? a.offsets[s] : 0.0f;
      dirty[u] = in ? a.sparse_index[(size_t)n * S + s] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < kMaxSlots; u++) {
      const int sl = u * kTmWaves + wave;
      if (sl >= NS) break;
      const int s = sl * 64 + lane;
      float4 v = sv[u];
      if (s < S) {
        v.w += off[u];
        if (!(v.w >= 0.0f)) v.w = qnan;
        if (dirty[u]) {
          reinterpret_cast<float4 *>(a.out_gradient)[(size_t)n * S + s] = make_float4(0.f, 0.f, 0.f, 0.f);
          a.sparse_index[(size_t)n * S + s] = 0;
        }
      }
      sph[s] = v;
      aterm[s] = 0.5f * (v.w * v.w - (v.x * v.x + v.y * v.y + v.z * v.z));
      const bool on = v.w == v.w;  // disabled / padding spheres: an empty box
      const float big = 3.0e38f;
      float e[6] = {on ? v.w - v.x : -big, on ? v.w - v.y : -big, on ? v.w - v.z : -big,
                    on ? v.x + v.w : -big, on ? v.y + v.w : -big, on ? v.z + v.w : -big};
#pragma unroll
      for (int c = 0; c < 6; c++) {
        e[c] = fmaxf(e[c], dpp_f<0xB1>(e[c]));
        e[c] = fmaxf(e[c], dpp_f<0x4E>(e[c]));
        e[c] = fmaxf(e[c], dpp_f<0x141>(e[c]));
        e[c] = fmaxf(e[c], dpp_f<0x140>(e[c]));
      }
      if (li == 0) {
        float *b = box + (sl * 4 + row) * kT2Box;
        b[0] = -e[0]; b[1] = -e[1]; b[2] = -e[2]; b[3] = e[3]; b[4] = e[4]; b[5] = e[5];
      }
    }
  }
  __syncthreads();
  // ---- level 1: this wavefront's share of the tiles whose block boxes overlap, compacted in list order
  int count = 0;
#pragma unroll
  for (int u = 0; u < kMaxTileTrips; u++) {
    if ((u * kTmWaves + wave) * kWave >= t.n_tiles) break;
    const int tl = my_tiles[u];
    bool keep = false;
    if (tl >= 0) {
      const float *bi = box + (tl & 0xff) * kT2Box, *bj = box + (tl >> 8) * kT2Box;
      keep = bi[0] <= bj[3] && bj[0] <= bi[3] && bi[1] <= bj[4] && bj[1] <= bi[4] && bi[2] <= bj[5] && bj[2] <= bi[5];
    }
    const unsigned long long m = __ballot(keep);
    if (keep) kept[count + __popcll(m & ((1ull << lane) - 1ull))] = tl | (((u * kTmWaves + wave) * kWave + lane) << 16);  // + its list index
    count += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- narrow phase: one tile per pair of matrix-core instructions.  Lane (k = lane / 16, m = lane % 16) feeds component k of
  // sphere 16 ib + m (A) and of sphere 16 jb + m (B); it receives rows 4 k .. 4 k + 3 of column m: the pairs
  // (16 ib + 4 k + reg, 16 jb + m).  HALF the penetration is accumulated: c_i . c_j + r_i r_j from the first instruction, the
  // row / column terms h_i + h_j (h = (r^2 - |c|^2) / 2) from a second one with A = (h_i, 1, 0, 0), B = (1, h_j, 0, 0) --
  // no accumulator initialisation, no doubling.  A disabled / padding sphere carries NaN: its whole row / column is NaN and
  // compares false.
  float bv = 0.0f;
  int bkey = 0x7fffffff;
  const int kk = lane >> 4, mm = lane & 15;
  const char *sphb = reinterpret_cast<const char *>(sph) + (mm * 4 + kk) * 4;  // + 256 * block: component kk of its sphere mm
  const char *ahb = reinterpret_cast<const char *>(aterm) + mm * 4;             // + 64 * block: h of its sphere mm
  const float sel0 = kk == 0 ? 1.0f : 0.0f, sel1 = kk == 1 ? 1.0f : 0.0f;
  const sc_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float thr = -0.5f * kMfmaSlack;
  constexpr int U = 4;  // tiles per trip: the trip's operand reads and matrix-core instructions are independent of each other
  for (int t0 = 0; t0 < count; t0 += U) {
    const int4 k4 = *reinterpret_cast<const int4 *>(kept + t0);  // (one LDS round trip per trip; kept_cap is a multiple of 64)
    const int kv[U] = {k4.x, k4.y, k4.z, k4.w};
    int tl[U];
    sc_f32x4 acc[U];
    uint32_t listed[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool in = t0 + u < count;
      tl[u] = __builtin_amdgcn_readfirstlane(in ? kv[u] : kv[0]);
      const int ib = tl[u] & 0xff, jb = (tl[u] >> 8) & 0xff;
      // which of this lane's four pairs are LISTED: requested with the operands (an L2 round trip that the matrix-core
      // instructions cover) -- most kept tiles hold close UNLISTED pairs (the spheres of one link, of neighbouring links), so
      // asking only after the close test would put the load on the dependent path of every other tile
      if (t.lane_masks != nullptr) {
        listed[u] = in ? (uint32_t)t.lane_masks[(size_t)(tl[u] >> 16) * 64 + lane] : 0u;
      } else {  // bit (16 (jb & 1) + m) of bitmap[jb / 2][i] <-> pair (i, 16 jb + m): the words of this lane's four rows i
        const uint4 w = *reinterpret_cast<const uint4 *>(a.bitmap + (size_t)(jb >> 1) * SL + ib * kTile + 4 * kk);
        const int sh = (jb & 1) * 16 + mm;
        listed[u] = in ? ((w.x >> sh) & 1u) | (((w.y >> sh) & 1u) << 1) | (((w.z >> sh) & 1u) << 2) | (((w.w >> sh) & 1u) << 3) : 0u;
      }
      const float av = *reinterpret_cast<const float *>(sphb + ib * 256);
      const float bw = *reinterpret_cast<const float *>(sphb + jb * 256);
      const float hi_ = *reinterpret_cast<const float *>(ahb + ib * 64);
      const float hj_ = *reinterpret_cast<const float *>(ahb + jb * 64);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, zero4, 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_fmaf(sel0, hi_, sel1), __builtin_fmaf(sel1, hj_, sel0), acc[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float m4 = fmaxf(fmaxf(fmaxf(acc[u][0], acc[u][1]), acc[u][2]), acc[u][3]);
      if (__ballot(m4 > thr && listed[u] != 0u) == 0ull) continue;  // no listed pair of this tile is close
      uint32_t cand = 0u;
#pragma unroll
      for (int reg = 0; reg < 4; reg++) cand |= (acc[u][reg] > thr ? 1u : 0u) << reg;
      cand &= listed[u];
      if (__ballot(cand != 0u) == 0ull) continue;  // (~30 listed pairs of a point penetrate)
      const int ib = tl[u] & 0xff, jb = (tl[u] >> 8) & 0xff;
      const int j = jb * kTile + mm;
      const float4 sj = sph[j];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        if ((cand >> reg) & 1u) {
          const int i = ib * kTile + 4 * kk + reg;
          const float v = pair_pen(sph[i], sj);
          const int key = (i << 10) | j;
          if (v > 0.0f && (v > bv || (v == bv && key < bkey))) { bv = v; bkey = key; }
        }
      }
    }
  }
  // ---- arg-max: largest penetration, then the lexicographically first (i, j); over the wavefront, then over the four
  float m = bv;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  int key = (m > 0.0f && bv == m) ? bkey : 0x7fffffff;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, kWave));
  if (lane == 0) {
    red[wave * 2] = m;
    reinterpret_cast<int *>(red)[wave * 2 + 1] = key;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  m = 0.0f;
  key = 0x7fffffff;
  for (int w = 0; w < kTmWaves; w++) {
    const float mw = red[w * 2];
    const int kw = reinterpret_cast<const int *>(red)[w * 2 + 1];
    if (mw > m || (mw == m && kw < key)) { m = mw; key = kw; }
  }
  if (!(m > 0.0f) || key == 0x7fffffff) {
    a.out_distance[n] = 0.0f;
    return;
  }
  const float wgt = a.weight[0];
  a.out_distance[n] = 0.5f * wgt * m;
  if (a.write_grad) {
    const int i = key >> 10, j = key & 1023;
    const float4 s1 = sph[i], s2 = sph[j];
    const float vx = wgt * (s2.x - s1.x), vy = wgt * (s2.y - s1.y), vz = wgt * (s2.z - s1.z);
    float4 *g = reinterpret_cast<float4 *>(a.out_gradient) + (size_t)n * S;
    g[i] = make_float4(vx, vy, vz, wgt * -1.0f);
    g[j] = make_float4(-1.0f * vx, -1.0f * vy, -1.0f * vz, wgt * -1.0f);
    a.sparse_index[(size_t)n * S + i] = 1;
    a.sparse_index[(size_t)n * S + j] = 1;
  }
}

template <int NWAVES>
static void launch_self(const SelfCollArgs &a, int blocks, size_t lds, int ppw, int tile, hipStream_t st) {
  if (a.store_pair_distance)
    hipLaunchKernelGGL((self_collision_kernel<NWAVES, true>), dim3(blocks), dim3(NWAVES * 64), lds, st, a, ppw, tile);
  else
    hipLaunchKernelGGL((self_collision_kernel<NWAVES, false>), dim3(blocks), dim3(NWAVES * 64), lds, st, a, ppw, tile);
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
  CUROBO_REQUIRE(nspheres >= 1 && nspheres <= 768, "%s: nspheres=%d out of range [1,768]", what, nspheres);
  CUROBO_REQUIRE(num_collision_pairs >= 0, "%s: negative num_collision_pairs", what);
  CUROBO_REQUIRE(((uintptr_t)pair_locations & 3) == 0, "%s: pair_locations must be 4-byte aligned", what);
  const long n_points = (long)batch_size * horizon;
  if (n_points == 0) return CUROBO_HIP_OK;  // (an empty batch has no buffers: nothing below may ask for them)
  CUROBO_REQUIRE(!store_pair_distance || pair_distance, "%s: store_pair_distance needs pair_distance", what);
  SelfCollArgs a{};
  a.out_distance = out_distance; a.out_gradient = out_vec; a.pair_distance = pair_distance;
  a.sparse_index = sparse_index; a.robot_spheres = robot_spheres; a.offsets = sphere_padding;
  a.weight = weight; a.pair_locations = pair_locations;
  a.n_points = (int)n_points; a.nspheres = nspheres; a.npairs = num_collision_pairs;
  a.store_pair_distance = store_pair_distance; a.write_grad = compute_grad;
  hipStream_t st = (hipStream_t)stream;
  const int P = num_collision_pairs;
  const size_t lds16 = (size_t)16 * nspheres * 16 + (size_t)(P > 0 ? P : 1) * 4;
  if (P <= 8192 && lds16 <= 60 * 1024) {
    // arms: 16 lanes per point, 16 points per workgroup, whole pair list resident in LDS
    const unsigned blocks = (unsigned)ceil_div_l(n_points, 16);
    if (store_pair_distance)
      hipLaunchKernelGGL((self_collision_row16_kernel<true>), dim3(blocks), dim3(256), lds16, st, a);
    else
      hipLaunchKernelGGL((self_collision_row16_kernel<false>), dim3(blocks), dim3(256), lds16, st, a);
  } else if (P <= 8192) {
    const size_t lds = (size_t)4 * nspheres * 16 + (size_t)(P > 0 ? P : 1) * 4;
    launch_self<4>(a, (int)ceil_div_l(n_points, 4), lds, 1, P > 0 ? P : 1, st);
  } else {
    // humanoids: 4096-pair tiles shared by 8 (or 4) concurrent points of a workgroup
    const int tile = 4096;
    const size_t lds8 = (size_t)8 * nspheres * 16 + (size_t)tile * 4;
    if (lds8 <= 64 * 1024) launch_self<8>(a, (int)ceil_div_l(n_points, 8), lds8, 1, tile, st);
    else launch_self<4>(a, (int)ceil_div_l(n_points, 4), (size_t)4 * nspheres * 16 + (size_t)tile * 4, 1, tile, st);
  }
  return check_launch(what, st);
}


CUROBO_EXPORT int curobo_hip_self_collision_distance_dense(
    float *out_distance, float *out_vec, uint8_t *sparse_index, const float *robot_spheres,
    const float *sphere_padding, const float *weight, const uint32_t *pair_bitmap, const int32_t *tile_list, int num_tiles,
    const uint8_t *tile_lane_masks, int batch_size, int horizon, int nspheres, int nslots, int compute_grad, curobo_hip_stream_t stream) {
  const char *what = "self_collision_distance_dense";
  CUROBO_REQUIRE(nspheres >= 1 && nspheres <= 1024, "%s: nspheres=%d out of range [1,1024]", what, nspheres);
  CUROBO_REQUIRE(nslots >= 1 && nslots * 64 >= nspheres && nslots % 4 == 0 && nslots <= 16,
                 "%s: nslots=%d must be a multiple of 4 with nslots * 64 >= nspheres (<= 16)", what, nslots);
  CUROBO_REQUIRE(pair_bitmap != nullptr, "%s: pair_bitmap is NULL", what);
  const long n_points = (long)batch_size * horizon;
  if (n_points == 0) return CUROBO_HIP_OK;
  SelfDenseArgs a{};
  a.out_distance = out_distance; a.out_gradient = out_vec; a.sparse_index = sparse_index;
  a.robot_spheres = robot_spheres; a.offsets = sphere_padding; a.weight = weight; a.bitmap = pair_bitmap;
  a.n_points = (int)n_points; a.nspheres = nspheres; a.nslots = nslots; a.write_grad = compute_grad;
  hipStream_t st = (hipStream_t)stream;
  static const bool no_tiles = getenv("CUROBO_HIP_SELF_NO_BROAD_PHASE") != nullptr;
  if (tile_list != nullptr && num_tiles > 0 && !no_tiles) {  // broad phase over 16 x 16 tiles
    static const bool one_level = getenv("CUROBO_HIP_SELF_ONE_LEVEL") != nullptr;  // (A/B knob: the round-2 kernel)
    SelfTilesArgs ta{a, tile_list, num_tiles, tile_lane_masks};
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(self_collision_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(self_collision_tiles2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr2 = true;
    }
    // The matrix-core narrow phase (self_collision_tiles_mfma_kernel) is bit-identical to the two-level kernel and, measured
    // on the G1 (profiles/r05_c_self_collision_mfma.json), exactly as fast: 7.3 M matrix-core instructions per launch replace
    // the second box level, but the launch stays bound by instruction issue of the code AROUND them (166 M VALU + 71 M scalar
    // wave-instructions, 4.1 SIMD-cycles each).  It runs on request (CUROBO_HIP_SELF_MFMA=1, the tests); the default stays the
    // two-level kernel.
    const bool use_mfma = getenv("CUROBO_HIP_SELF_MFMA") != nullptr;  // (read per call: the tests switch it)
    if (!one_level && use_mfma) {  // level-1 block boxes, then one 16 x 16 tile per pair of matrix-core instructions
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(self_collision_tiles_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr3 = true;
      }
      const size_t lds = tilesm_lds_floats(nslots, num_tiles) * sizeof(float);
      CUROBO_REQUIRE(lds <= 64 * 1024 && num_tiles <= 2048 && nslots <= 16, "%s: too many tiles / spheres for the LDS tiling", what);
      hipLaunchKernelGGL(self_collision_tiles_mfma_kernel, dim3((unsigned)n_points), dim3(kTmWaves * 64), lds, st, ta);
      return check_launch(what, st);
    }
    if (!one_level) {  // two-level broad phase, four wavefronts per point
      const size_t lds = tiles2_lds_floats(nslots, num_tiles) * sizeof(float);
      CUROBO_REQUIRE(lds <= 64 * 1024 && num_tiles <= 2048 && nslots <= 16, "%s: too many tiles / spheres for the LDS tiling", what);
      hipLaunchKernelGGL(self_collision_tiles2_kernel, dim3((unsigned)n_points), dim3(kT2Waves * 64), lds, st, ta);
      return check_launch(what, st);
    }
    const size_t wave_bytes = ((size_t)nslots * 64 * 4 + (size_t)nslots * 4 * 8 + (size_t)((num_tiles + 3) & ~3)) * sizeof(float);
    // one wavefront per workgroup for big robots: the CU then holds as many wavefronts as its LDS allows, not a
    // multiple of a workgroup's
    const int wv = wave_bytes >= 8 * 1024 ? 1 : 4;
    CUROBO_REQUIRE(wave_bytes * wv <= 64 * 1024, "%s: too many tiles / spheres for the LDS tiling", what);
    hipLaunchKernelGGL(self_collision_tiles_kernel, dim3((unsigned)ceil_div_l(n_points, wv)), dim3(wv * 64), wave_bytes * wv, st, ta);
    return check_launch(what, st);
  }
  const size_t lds_wave = (size_t)nslots * 64 * 16;
  const int waves = lds_wave * 4 <= 64 * 1024 ? 4 : (lds_wave * 2 <= 64 * 1024 ? 2 : 1);
  const size_t lds = lds_wave * waves;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(self_collision_dense_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((self_collision_dense_kernel<4>), dim3((unsigned)ceil_div_l(n_points, waves)), dim3(waves * 64), lds, st, a);
  return check_launch(what, st);
}
