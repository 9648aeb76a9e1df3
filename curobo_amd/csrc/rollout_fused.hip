// rollout_fused.hip -- the whole rollout of one trajectory in ONE kernel launch:
//   knots -> B-spline -> FK -> collision spheres -> self collision + (swept) scene collision
//         -> per-trajectory cost, and the VJP back through FK and the B-spline to the knots.
//
// This is the MI355X-first form of the hot path.  The reference (and the drop-in entry points of
// this library) run 7 kernels that hand ~6.8 KB per trajectory point through HBM (joint angles,
// 13 cumulative transforms, 65 spheres, two 65x4 gradient buffers, ...).  None of those tensors
// is consumed by the optimiser: L-BFGS only needs cost[B] and d cost / d knots.  Here a
// workgroup owns one trajectory; every intermediate lives in LDS (2.3 KB per point for a Franka,
// two workgroups per CU), HBM traffic drops to ~1.8 KB per ROLLOUT (measured: knots in, cost +
// gradient out, tables from L2), and the six launch boundaries disappear.  Optional pointers materialise joint positions
// and world spheres for callers that want them (metrics / visualisation).
//
// Arithmetic is shared with the stand-alone kernels through the *_device.hpp headers, so the
// fused and unfused paths agree to fp32 summation order (tests/test_gpu_fused.py).
//
// Mapping: a trajectory point is owned by a 16-lane DPP row exactly as in kinematics.hip
// (4 points per wave64).  Phases (workgroup barriers between them):
//   P0  all lanes: stage robot tables, pair list, obstacle records; B-spline samples -> q in LDS
//   P1  per point: local transforms (one sincos per lane) -> barrier-free chain -> spheres
//   P2  self collision per row (DPP arg-max over the padded pair list); per-link obstacle masks; the
//       scene pass of a wave packs its rows' (sphere, obstacle) pairs into LDS rings and evaluates
//       them 64 at a time (wave_scene_pass); sphere gradients go into per-link wrenches in a fixed
//       order; then, per point, every moving link gathers the wrenches of its subtree -> grad_q.
//       Rows of a wave take points strided along the trajectory; a 33rd ("leftover") point is shared
//       by the whole workgroup.  Optional passes (TERMS): tool pose, c-space STATE.
//   P3  B-spline VJP to the knots (four streams with TERMS), fixed-order sum of the point costs
// Workgroups take their trajectory through an optional longest-first permutation that the previous
// launches built from measured workgroup durations (rebuild_dispatch_order).  DESIGN.md section 4.1
// has the measurements behind each of these choices.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <vector>

#include "bspline_device.hpp"
#include "cost_device.hpp"
#include "fused_shapes.hpp"
#include "dynamics_device.hpp"
#include "fk_device.hpp"
#include "scene_device.hpp"
#include "self_device.hpp"

// 0: the main translation unit (generic kernels + the C ABI); k > 0: ONLY the instantiations of compile-time shape k and their
// launcher (fused_shapes.hpp; curobo_amd/build.py compiles this file once per shape, in parallel)
#ifndef CUROBO_FUSED_SHAPE_TU
#define CUROBO_FUSED_SHAPE_TU 0
#endif

namespace curobo_hip {

struct FusedTrajArgs {
  float *out_cost;        // [B]
  float *out_grad_knots;  // [B, n_knots, D]
  float *out_position;    // optional [B, H, D]
  float *out_spheres;     // optional [B, H, S, 4]
  BsFwdArgs bs;           // knots + start/goal states + dt (out_* members unused)
  const float *fixed_transform, *robot_spheres, *joint_offset;
  const int8_t *joint_map_type;
  const int16_t *joint_map, *link_map, *link_sphere_map, *link_chain_data, *link_chain_offsets;
  const float *sphere_padding, *w_self;
  const int16_t *pairs;
  const uint32_t *lane_lists;  // optional: the pair list re-ordered for lane = sphere (curobo_hip_self_lane_lists_host)
  int lane_len0, lane_len1;    // list entries per lane of pass 0 (spheres 0..63) and pass 1 (64..127)
  curobo_hip_scene sc;
  const float *w_scene, *eta, *speed_dt;
  const int32_t *env_query_idx;
  int batch, nlinks, nspheres, npairs, chain_len, dpad, num_envs, use_multi_env, enable_speed_metric;
  int use_self, use_scene;
  // optional cost terms of the full trajopt task (lbfgs_bspline_trajopt.yml): tool-pose goal cost
  // over the horizon (terminal / non-terminal weights) and the c-space STATE cost
  ToolPoseArgs tp;        // current_position / current_quat unused; out_* optional [B, H, T, .]
  CspaceStateArgs cs;     // pos/vel/acc/jerk unused (LDS); out_cost optional [B, H, D]; out_g* unused
  const int16_t *tool_frame_map;
  int n_tool_frames, use_pose, use_cspace;
  // optional joint-torque limits (c-space STATE effort terms on tau = RNEA(q, qd, qdd)): inverse dynamics and its VJP
  // run inside the launch on LDS regions that are dead by then (see fused_torque_fits)
  const float *link_masses_com, *link_inertias, *gravity;
  const int16_t *level_links;
  int use_torque;
  // optional longest-first dispatch (see rebuild_dispatch_order): int32 [4][B] = order[2][B], ticks[2][B]
  int32_t *dispatch_ws;
  int dispatch_phase;
  int scene_rows;  // development knob: one sphere per lane in the scene pass (CUROBO_HIP_SCENE_ROWS)
  long long *prof;  // optional [B][16] wall-clock ticks (100 MHz) at the phase boundaries, see set_profile_buffer
};

// LDS carve (floats unless noted), per workgroup:
//   q / grad_q [H][D] | cumul [H][L][12] | work [H][WS] (locals [L][16] then spheres [S][4])
//   | wrench [H][L][7] | cost [H] | parent[L] chain_off[L+1] link_info[L] sign[L] chain[C]
//   offset_add[L] fixed_transform[L][12] sphere_link[S] sphere_rad[S] (raw radius) sphere_pad[S]
//   link-frame spheres [S][4] | link bounding boxes [L][8] (ordered-int keys) | subtree masks [L][4] | joint-link masks [D][4]
//   | leftover-point sphere gradients [S][4] + arg-max key | pairs [P] | obstacle records
constexpr int kWrench = 7;  // per link: force xyz, torque xyz about the link origin, joint gradient
constexpr int kSceneListEntries = 128;  // ring of active (row, sphere) entries per wave: < 64 pending + <= 64 appended

struct FusedLayout {
  int q, cumul, work, ws, wrench, wl, cost, parent, chain_off, link_info, sign, lists, off_add, fixed, chain, sph_link, sph_rad,
      sph_pad, rs, lbound, sub, jlinks, left, key, flag, dyn, cstab, pairs, lanel, recs, total;
};
// n_lane > 0: the lane = sphere form of the pair list (n_lane words) is staged INSTEAD of the (i, j) offsets
__host__ __device__ inline FusedLayout fused_layout(int H, int D, int L, int S, int C, int P, int n_rec, int n_dyn = 0,
                                                    int n_waves = 0, int rings = 1, int n_lane = 0, bool with_left = true) {
  FusedLayout f;
  int o = 0;
  auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };  // 16-byte granules
  f.q = take(H * D);
  f.cumul = take(H * L * 12);
  f.ws = ((L * 16 > (S + 1) * 4 ? L * 16 : (S + 1) * 4) + 3) & ~3;  // + one all-NaN sphere behind the last one
  f.work = take(H * f.ws);
  f.wl = L * kWrench;
  f.wrench = take(H * f.wl);
  f.cost = take(H);
  f.parent = take(L);
  f.chain_off = take(L + 1);
  f.link_info = take(L);
  f.sign = take(L);
  // tables that are dead after P1 (staging + FK) share their bytes with the per-wave lists of active
  // scene spheres of P2 (`rings` x kSceneListEntries uint16 per wave)
  f.lists = o;
  f.off_add = take(L);
  f.fixed = take(L * 12);
  f.chain = take(C);
  f.sph_pad = take(S);
  f.rs = take(S * 4);
  if (o - f.lists < n_waves * rings * kSceneListEntries / 2) o = f.lists + n_waves * rings * kSceneListEntries / 2;
  f.sph_link = take(S);
  f.sph_rad = take(S);
  f.lbound = take(L * 8);
  f.sub = take(L * 4);
  f.jlinks = take(D * 4);
  f.left = take(with_left ? S * 4 : 0);  // (the trajectory kernel's leftover-point buffer: the IK launch has none, and its 1 KB is
                                         // the difference between three and four workgroups per CU there)
  f.key = take(4);
  f.flag = take(H);  // per point: any wrench written
  f.dyn = take(n_dyn);  // velocity / acceleration / jerk (+ joint-space position gradient) [4][H][D] when the c-space STATE cost is on
  f.cstab = take(n_dyn ? 10 * D + 12 : 0);  // c-space limits (shrunk) [10][D] (pos, vel, acc, jerk, effort) + 10 retimed weights + dt
  f.pairs = take(n_lane > 0 ? 0 : (P + 63) & ~63);  // padded with (NaN sphere, NaN sphere) pairs: loops need no bounds checks
  f.lanel = take(n_lane);
  f.recs = take(n_rec * kObsRecFloats);
  f.total = o;
  return f;
}

// Inverse dynamics inside the launch (torque limits) borrows LDS that is dead when it runs:
//   link constants [L][24 + 4]        <- the P1-only tables / scene rings (lists .. sph_link)
//   q, qd, qdd copies [3][H][D]       <- the pair list (dead after the collision pass)
//   tau [H][D]                        <- the leftover-point buffer `left`
//   d cost / d tau [H][D]             <- sph_link + sph_rad + lbound
//   forward cache [L][20][H]          <- the sphere rows `work` (dead after the collision pass)
//   adjoints f, a [2][L][6][H]        <- cumul (dead after the wrench gather: exactly H L 12 floats)
//   adjoint v [L][6][H]               <- wrench (dead after the gather)
__host__ __device__ inline bool fused_torque_fits(const FusedLayout &f, int H, int D, int L, int S) {
  return L * (kLinkFloats + 4) <= f.sph_link - f.lists && 3 * H * D <= f.recs - f.pairs && H * D <= S * 4 &&
         H * D <= f.sub - f.sph_link && L * 20 <= f.ws && L * 6 <= L * kWrench;
}

__device__ __forceinline__ float uniform_f(float v) {  // wave-uniform value -> SGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// LDS views + per-launch scalars shared by the phases
struct FusedCtx {
  float *q, *cumul, *work, *wrench, *cost, *sign, *off_add, *fixed, *sph_rad, *sph_pad;
  float4 *rs;            // link-frame spheres of this workgroup's robot instance
  int *lbound;           // per link: box (link frame) around its collision spheres: lo xyz, hi xyz as ordered-int keys
  int *parent, *chain_off, *link_info, *chain, *sph_link;
  uint32_t *sub, *jlinks, *pairs;
  const uint32_t *lanel;  // [lane_len0 + lane_len1][64]: partner byte offset | pair index << 16
  const uint32_t *g_pairs;  // the (i, j) list in global memory (read for the one winning pair when lanel is in use)
  int lane_len0, lane_len1;
  float4 *left;
  uint16_t *lists;  // [waves][kSceneListEntries], overlays the P1-only tables
  int *flag;   // [H] point has gradients (set by the cost pass, read by the VJP pass)
  float *dyn;  // [3][H][D] velocity / acceleration / jerk, later their cost gradients
  float *cstab;  // c-space STATE constants staged once per workgroup: limits [8][D], weights [10], dt
  unsigned long long *key;
  ObsRec *recs;
  int H, D, L, S, P, ws, wl, env;
  float w_self, w_scene, eta, speed_dt;
  bool speed_metric;
  __device__ __forceinline__ const float4 *spheres(int h) const { return reinterpret_cast<const float4 *>(work + (size_t)h * ws); }
};

// Adds the cost gradient g acting at world point p of a sphere on link l to that link's wrench
// accumulator (force, torque about the link origin).  Called by ONE lane at a time (the callers
// serialise the contributing lanes in lane order), so the fp32 sums are reproducible.
__device__ __forceinline__ void wrench_add(float *__restrict__ wr, const float *__restrict__ cumul, int l, f3 p, f3 g) {
  const float *C = cumul + l * 12;
  const f3 t = cross(p - make_f3(C[3], C[7], C[11]), g);
  float *w = wr + l * kWrench;
  atomicAdd(w + 0, g.x); atomicAdd(w + 1, g.y); atomicAdd(w + 2, g.z);  // ds_add_f32, fire and forget
  atomicAdd(w + 3, t.x); atomicAdd(w + 4, t.y); atomicAdd(w + 5, t.z);
}

// the lanes of the wave whose sphere gradient is non-zero add their wrench one at a time, in lane
// order; returns whether the caller's 16-lane row had any.  Everything that needs a wait (link
// origin from LDS, the torque) is computed by all lanes before the serial section: that section is
// the tail of points deep in collision (one turn per contributing sphere), so it only issues the
// six ds_add_f32.
__device__ __forceinline__ bool wrench_add_serialised(const FusedCtx &c, int h, int s, f3 p, f3 g, int lane64) {
  unsigned long long m = __ballot(g.x != 0.0f || g.y != 0.0f || g.z != 0.0f);
  const bool row_any = ((m >> (lane64 & 48)) & 0xffffull) != 0ull;
  if (m) {
    const int l = c.sph_link[s < c.S ? s : 0];
    const float *C = c.cumul + (size_t)h * c.L * 12 + l * 12;
    const f3 t = cross(p - make_f3(C[3], C[7], C[11]), g);
    float *w = c.wrench + (size_t)h * c.wl + l * kWrench;
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (lane64 == src) {
        atomicAdd(w + 0, g.x); atomicAdd(w + 1, g.y); atomicAdd(w + 2, g.z);  // ds_add_f32, fire and forget
        atomicAdd(w + 3, t.x); atomicAdd(w + 4, t.y); atomicAdd(w + 5, t.z);
      }
    }
  }
  return row_any;
}

// squared-distance penetration of one staged pair (reference sphere_squared_distance_fused,
// self_collision_helper.cuh:61-71); ij = byte offsets of the two spheres, NaN when either is disabled
__device__ __forceinline__ float sphere_pair_penetration(float4 s1, float4 s2) {  // the same bits whichever sphere comes first
  const float r = s1.w + s2.w;
  const float dx = s1.x - s2.x, dy = s1.y - s2.y, dz = s1.z - s2.z;
  return (r * r) - (dx * dx + dy * dy + dz * dz);
}
__device__ __forceinline__ float staged_pair_penetration(const float4 *sph, uint32_t ij) {
  const float4 s1 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sph) + (ij & 0xffffu));
  const float4 s2 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sph) + (ij >> 16));
  return sphere_pair_penetration(s1, s2);
}
// max of two values that are never signalling NaNs (v_max_f32 returns the other operand for a quiet NaN): without the
// v_max x, x canonicalisation the compiler puts in front of every fmaxf whose operand it cannot prove quiet
__device__ __forceinline__ float max_quiet(float a, float b) {
  float o;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
// wave64 reductions to a scalar: the row, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lane 63 holds
// the result (no LDS round trip, the result is in an SGPR: branches on it are scalar)
__device__ __forceinline__ float wave64_max(float v) {
  v = row16_max(v);
  int x = __builtin_bit_cast(int, v);
  x = __builtin_bit_cast(int, fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false))));
  v = __builtin_bit_cast(float, x);
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false)));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave64_min(int v) {
  v = row16_min(v);
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned long long pair_key(float pen, int k) {  // max = largest pen, then lowest k
  return ((unsigned long long)__float_as_uint(pen) << 32) | (unsigned long long)(0x7fffffffu - (uint32_t)k);
}

// the arg-max pair pushes its two spheres apart (reference self_collision_kernel.cuh:84-111)
__device__ __forceinline__ float self_pair_apply(const FusedCtx &c, int h, float m, int k) {
  const float4 *sph = c.spheres(h);
  const uint32_t ij = c.g_pairs ? (c.g_pairs[k] << 4) : c.pairs[k];
  const int i = (int)((ij & 0xffffu) >> 4), j = (int)(ij >> 20);
  const float4 s1 = sph[i], s2 = sph[j];
  const f3 g = make_f3(c.w_self * (s2.x - s1.x), c.w_self * (s2.y - s1.y), c.w_self * (s2.z - s1.z));
  float *wr = c.wrench + (size_t)h * c.wl;
  const float *cumul = c.cumul + (size_t)h * c.L * 12;
  wrench_add(wr, cumul, c.sph_link[i], make_f3(s1.x, s1.y, s1.z), g);
  wrench_add(wr, cumul, c.sph_link[j], make_f3(s2.x, s2.y, s2.z), -1.0f * g);
  return 0.5f * c.w_self * m;
}

// order-preserving float <-> int map (its own inverse) so that integer atomic min/max order floats
__device__ __forceinline__ int float_key(float f) { const int k = __float_as_int(f); return k >= 0 ? k : k ^ 0x7fffffff; }
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// Obstacle mask of every link of point h (lanes = links): the link's bounding ball against the
// obstacles, with the sweep reach bounded through the link's own motion,
//   |c_s(h+-1) - c_s(h)| <= |C(h+-1) - C(h)| + ||R(h+-1) - R(h)||_F |x_s - x_centre|.
// The mask is parked in the link's (not yet used) joint-gradient slot of the wrench table.
template <int SWEEP, int KINDS>
__device__ __forceinline__ void point_link_masks(const FusedCtx &c, const curobo_hip_scene &sc, int h, int lane) {
  float *wr = c.wrench + (size_t)h * c.wl;
  for (int l = lane; l < c.L; l += kFkLanes) {
    // ball around the link's box of collision spheres: centre, half diagonal (+ rounding margin)
    const int *bx = c.lbound + l * 8;
    const f3 lo = make_f3(key_float(bx[0]), key_float(bx[1]), key_float(bx[2]));
    const f3 hi = make_f3(key_float(bx[4]), key_float(bx[5]), key_float(bx[6]));
    const f3 hd = 0.5f * (hi - lo);
    const float4 lb = make_float4(0.5f * (hi.x + lo.x), 0.5f * (hi.y + lo.y), 0.5f * (hi.z + lo.z),
                                  lo.x <= hi.x ? sqrtf(dot(hd, hd)) * 1.0001f + 1e-6f : -1.0f);
    uint32_t mask = 0u;
    if (lb.w >= 0.0f) {
      const float *M = c.cumul + ((size_t)h * c.L + l) * 12;
      const float4 C4 = transform_sphere(M, lb);
      const f3 C = make_f3(C4.x, C4.y, C4.z);
      float reach = 0.0f;
      if (SWEEP > 0) {
#pragma unroll
        for (int dir = 0; dir < 2; dir++) {
          const int hn = dir == 0 ? h - 1 : h + 1;
          if (hn >= 0 && hn < c.H) {
            const float *N = c.cumul + ((size_t)hn * c.L + l) * 12;
            const float4 Cn = transform_sphere(N, lb);
            const f3 dC = make_f3(Cn.x - C.x, Cn.y - C.y, Cn.z - C.z);
            float fr = 0.0f;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int k = 0; k < 3; k++) { const float dr = N[r * 4 + k] - M[r * 4 + k]; fr += dr * dr; }
            reach = fmaxf(reach, 0.5f * (sqrtf(dot(dC, dC)) + sqrtf(fr) * lb.w));
          }
        }
        reach = reach * 1.001f + 1e-5f;
      }
      mask = bounding_ball_obstacle_mask<KINDS>(sc, c.recs, C, lb.w, c.eta, reach);
    }
    wr[l * kWrench + 6] = __uint_as_float(mask);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// scene cost + gradient of sphere s of point h (neighbour spheres from LDS for the sweep / speed metric)
template <int SWEEP, int KINDS>
__device__ __forceinline__ float4 scene_sphere(const FusedCtx &c, const curobo_hip_scene &sc, int h, int s, float &d, f3 &g,
                                               uint32_t mask = 0xffffffffu) {
  const bool need_nb = SWEEP > 0 || c.speed_metric;
  const bool has_prev = need_nb && h > 0, has_next = need_nb && h < c.H - 1;
  float4 c4 = c.spheres(h)[s];
  c4.w = c.sph_rad[s];  // scene collision uses the raw radius
  sphere_scene_cost<SWEEP, true, KINDS>(sc, c.recs, c.env, c4, has_prev, c.spheres(h > 0 ? h - 1 : h)[s], has_next,
                                        c.spheres(h < c.H - 1 ? h + 1 : h)[s], c.eta, c.w_scene, c.speed_metric, c.speed_dt, d, g, mask);
  return c4;
}

// OR over the 64 lanes of the wave (uniform result)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  int x = (int)v;
  x |= dpp_i<0xB1>(x);
  x |= dpp_i<0x4E>(x);
  x |= dpp_i<0x141>(x);
  x |= dpp_i<0x140>(x);
  return (uint32_t)(__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) | __builtin_amdgcn_readlane(x, 32) |
                    __builtin_amdgcn_readlane(x, 48));
}

// Scene pass of the (up to) four points of a wave.  Only spheres on links whose bounding ball
// reaches an obstacle's activation shell do any work (a quarter of them on the C2 workload), few of
// those get past the per-obstacle early reject, and the ones that do cost up to 1 + 2 * SWEEP
// signed-distance evaluations per obstacle: with one sphere per lane, a row waits for its one lane
// that sweeps through several obstacles.  The unit of work is therefore a (sphere, obstacle) pair
// that passed the early reject.  The wave packs those of its rows into a ring in LDS (ballot + mbcnt
// compaction) and evaluates them 64 at a time, whichever row they belong to.  Costs and link
// wrenches are then added by one lane at a time in ring order (per point: sphere block, obstacle,
// sphere), so the fp32 sums are reproducible.  The speed metric is linear in (cost, gradient) and is
// applied per pair.  Entry = sphere (9 bits) | row (2) | obstacle record (5): S <= 512, <= 32 records.
constexpr int kScenePassMaxSpheres = 512, kScenePassMaxRecords = 32;
template <int SWEEP, int KINDS, bool DENSE>
__device__ __forceinline__ void wave_scene_pass(const FusedCtx &c, const curobo_hip_scene &sc, int h, bool valid, int lane,
                                                int lane64, uint16_t *ring, int row_stride, int fill_to = 64) {
  const int row = lane64 >> 4;
  const int n_rec = sc.max_cuboids + sc.max_voxel_grids;
  const bool need_nb = SWEEP > 0 || c.speed_metric;
  const float *wr = c.wrench + (size_t)(valid ? h : 0) * c.wl;
  // geometry of sphere s of point hh shared by the reject test and the evaluation
  struct Geo { f3 center, pp, np; float r_adj, half_prev, half_next; bool has_prev, has_next, enabled = false; };
  auto geometry = [&](int hh, int s) {
    Geo q;
    const float4 c4 = c.spheres(hh)[s];
    const float r = c.sph_rad[s];  // scene collision uses the raw radius
    q.enabled = r >= 0.0f;
    q.center = make_f3(c4.x, c4.y, c4.z);
    q.r_adj = r + c.eta;
    q.has_prev = need_nb && hh > 0;
    q.has_next = need_nb && hh < c.H - 1;
    const float4 p4 = c.spheres(hh > 0 ? hh - 1 : hh)[s], n4 = c.spheres(hh < c.H - 1 ? hh + 1 : hh)[s];
    q.pp = make_f3(p4.x, p4.y, p4.z);
    q.np = make_f3(n4.x, n4.y, n4.z);
    q.half_prev = q.half_next = 0.0f;
    if (SWEEP > 0) {
      if (q.has_prev) { const f3 dd = q.pp - q.center; q.half_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (q.has_next) { const f3 dd = q.np - q.center; q.half_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    return q;
  };
  int head = 0, pending = 0, s0 = 0;  // uniform: ring state, cursor over the sphere blocks ...
  uint32_t done = 0u;                 // ... and the records of block s0 already looked at
  // DENSE: a second ring (behind the first) packs the spheres whose link mask is not empty, so that the
  // obstacle tests below run on full wavefronts: a lane then walks the set bits of ITS sphere's mask
  // (with one sphere per lane and one obstacle per step, 7 of 8 lanes idled through the tests).
  uint16_t *ring_a = ring + kSceneListEntries;
  int head_a = 0, pend_a = 0;
  uint32_t bits = 0u;   // per lane: obstacle bits of my packed sphere still to test
  unsigned mine = 0u;   // per lane: my packed sphere (s | row << 9)
  for (;;) {
    if (DENSE) {
      while (pending < fill_to) {
        if (__ballot(bits != 0u) == 0ull) {  // the packed batch is used up: pack the next one
          while (pend_a < 64 && s0 < c.S) {
            const int s = s0 + lane;
            const bool has = valid && s < c.S && __float_as_uint(wr[c.sph_link[s < c.S ? s : 0] * kWrench + 6]) != 0u;
            const unsigned long long ball = __ballot(has);
            if (has) {
              const int at = head_a + pend_a + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ball >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ball, 0u));
              ring_a[at & (kSceneListEntries - 1)] = (uint16_t)(s | (row << 9));
            }
            pend_a += __builtin_popcountll(ball);
            s0 += kFkLanes;
          }
          if (pend_a == 0) break;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int cnt = pend_a < 64 ? pend_a : 64;
          bits = 0u;
          if (lane64 < cnt) {
            mine = ring_a[(head_a + lane64) & (kSceneListEntries - 1)];
            const int sm = (int)(mine & 511u), hm = h + ((int)(mine >> 9) - row) * row_stride;
            bits = __float_as_uint(c.wrench[(size_t)hm * c.wl + c.sph_link[sm] * kWrench + 6]);
            if (n_rec < 32) bits &= (1u << n_rec) - 1u;
          }
          head_a += cnt;
          pend_a -= cnt;
        }
        // geometry of my sphere once per visit (it is recomputed after an evaluation round in between)
        Geo q;
        float reach = 0.0f, thr2 = 0.0f;
        const int sm = (int)(mine & 511u), rm = (int)(mine >> 9);
        if (bits != 0u) {
          q = geometry(h + (rm - row) * row_stride, sm);
          reach = SWEEP > 0 ? fmaxf(q.half_prev, q.half_next) * 1.0001f + 2e-6f : 2e-6f;
          thr2 = (q.r_adj + reach) * (q.r_adj + reach) * 1.00001f;
        }
        do {
          bool pass = false;
          int j = 0;
          if (bits != 0u) {
            j = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const ObsRec rec = c.recs[j];
            if (rec.meta.x != 0.0f && q.enabled) {
              const f3 lc = to_local(rec, q.center);
              const bool vox = (KINDS & 2) && (!(KINDS & 1) || j >= sc.max_cuboids);
              pass = vox ? !obstacle_early_reject<true>(sc, rec, lc, q.r_adj, reach, thr2)
                         : !obstacle_early_reject<false>(sc, rec, lc, q.r_adj, reach, thr2);
            }
          }
          const unsigned long long ball = __ballot(pass);
          if (pass) {
            const int at = head + pending + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ball >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ball, 0u));
            ring[at & (kSceneListEntries - 1)] = (uint16_t)(sm | (rm << 9) | (j << 11));
          }
          pending += __builtin_popcountll(ball);
        } while (pending < fill_to && __ballot(bits != 0u) != 0ull);
      }
    } else {
    while (pending < fill_to && s0 < c.S) {  // fill
      const int s = s0 + lane;
      const bool in = valid && s < c.S;
      const uint32_t lmask = in ? __float_as_uint(wr[c.sph_link[s] * kWrench + 6]) : 0u;
      uint32_t todo = wave_or(lmask) & ~done;  // records some lane of the wave still has to test
      if (todo != 0u) {
        Geo q;
        float reach = 0.0f, thr2 = 0.0f;
        if (lmask != 0u) {
          q = geometry(h, s);
          reach = SWEEP > 0 ? fmaxf(q.half_prev, q.half_next) * 1.0001f + 2e-6f : 2e-6f;
          thr2 = (q.r_adj + reach) * (q.r_adj + reach) * 1.00001f;
        }
        while (todo != 0u && pending < fill_to) {
          const int j = __ffs((int)todo) - 1;
          todo &= todo - 1u;
          done |= 1u << j;
          bool pass = false;
          if (((lmask >> j) & 1u) && q.enabled) {
            const ObsRec rec = c.recs[j];
            if (rec.meta.x != 0.0f) {
              const f3 lc = to_local(rec, q.center);
              const bool vox = (KINDS & 2) && (!(KINDS & 1) || j >= sc.max_cuboids);
              pass = vox ? !obstacle_early_reject<true>(sc, rec, lc, q.r_adj, reach, thr2)
                         : !obstacle_early_reject<false>(sc, rec, lc, q.r_adj, reach, thr2);
            }
          }
          const unsigned long long ball = __ballot(pass);
          if (pass) {
            const int at = head + pending + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ball >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ball, 0u));
            ring[at & (kSceneListEntries - 1)] = (uint16_t)(s | (row << 9) | (j << 11));
          }
          pending += __builtin_popcountll(ball);
        }
      }
      if (todo == 0u) { s0 += kFkLanes; done = 0u; }
    }
    }
    if (pending == 0) break;
    const int count = pending < 64 ? pending : 64;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float d = 0.0f;
    f3 g = make_f3(0.f, 0.f, 0.f), center = g;
    int he = 0, se = 0;
    if (lane64 < count) {
      const unsigned e = ring[(head + lane64) & (kSceneListEntries - 1)];
      se = (int)(e & 511u);
      he = h + ((int)((e >> 9) & 3u) - row) * row_stride;
      const int je = (int)(e >> 11);
      const ObsRec rec = c.recs[je];
      const Geo q = geometry(he, se);
      center = q.center;
      const f3 lc = to_local(rec, q.center);
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      const bool vox = (KINDS & 2) && (!(KINDS & 1) || je >= sc.max_cuboids);
      if (vox)
        obstacle_contribution<true, SWEEP>(sc, rec, c.env * sc.max_voxel_grids + je - sc.max_cuboids, lc, q.has_prev, q.has_next,
                                           q.pp, q.np, q.r_adj, c.eta, q.half_prev, q.half_next, cost_sum, grad_local);
      else
        obstacle_contribution<false, SWEEP, (KINDS & 4) != 0>(sc, rec, c.env * sc.max_cuboids + je, lc, q.has_prev, q.has_next, q.pp, q.np, q.r_adj,
                                            c.eta, q.half_prev, q.half_next, cost_sum, grad_local);
      if (cost_sum > 0.0f) {
        d = c.w_scene * cost_sum;
        g = c.w_scene * to_world_vector(rec, grad_local);
        if (c.speed_metric && q.has_prev && q.has_next) speed_metric_apply(q.center, q.pp, q.np, c.speed_dt, d, g);
      }
    }
    unsigned long long m = __ballot(g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || d != 0.0f);
    if (m) {  // waits (link origin, torque) before the serial section, which only issues LDS atomics
      const int l = c.sph_link[se];
      const float *C = c.cumul + (size_t)he * c.L * 12 + l * 12;
      const f3 t = cross(center - make_f3(C[3], C[7], C[11]), g);
      float *w = c.wrench + (size_t)he * c.wl + l * kWrench;
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (lane64 == src) {
          atomicAdd(w + 0, g.x); atomicAdd(w + 1, g.y); atomicAdd(w + 2, g.z);
          atomicAdd(w + 3, t.x); atomicAdd(w + 4, t.y); atomicAdd(w + 5, t.z);
          atomicAdd(&c.cost[he], d);
          c.flag[he] = 1;
        }
      }
    }
    head += count;
    pending -= count;
  }
}

// Second half of the VJP of point h, run by its 16-lane row after all wrenches are in: every
// moving link sums the wrenches of its subtree about its own origin and projects them on its joint
// axis; then every dof sums its links (mimic joints) in a fixed order.  The reference walks the
// chain once per sphere (kinematics_backward_helper.cuh:62-98, kinematics_joint_util.cuh:13-66);
// the sum is the same, factored through the link wrenches.
__device__ __forceinline__ void point_vjp_gather(const FusedCtx &c, int h, bool any_grad, int lane) {
  const float *cumul = c.cumul + (size_t)h * c.L * 12;
  float *wr = c.wrench + (size_t)h * c.wl;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int l = lane; l < c.L; l += kFkLanes) {
    float r = 0.0f;
    const int info = c.link_info[l];
    const int jt = (info & 0xff) - 1;
    if (any_grad && jt >= J_X_PRISM) {
      const float *C = cumul + l * 12;
      const f3 o = make_f3(C[3], C[7], C[11]);
      f3 F = make_f3(0.f, 0.f, 0.f), T = make_f3(0.f, 0.f, 0.f);
      for (int wd = 0; wd < (c.L + 31) / 32; wd++) {
        uint32_t mask = c.sub[l * 4 + wd];
        while (mask) {
          const int lp = wd * 32 + __ffs((int)mask) - 1;
          mask &= mask - 1;
          const float *w = wr + lp * kWrench;
          const f3 f = make_f3(w[0], w[1], w[2]);
          const float *Cp = cumul + lp * 12;
          F = F + f;
          T = T + make_f3(w[3], w[4], w[5]) + cross(make_f3(Cp[3], Cp[7], Cp[11]) - o, f);
        }
      }
      const int ax = jt >= J_X_ROT ? jt - J_X_ROT : jt;
      const f3 axis = make_f3(C[ax], C[4 + ax], C[8 + ax]);
      r = c.sign[l] * (jt >= J_X_ROT ? dot(axis, T) : dot(axis, F));
    }
    wr[l * kWrench + 6] = r;  // slot 6 of link l: its joint gradient
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < c.D; d += kFkLanes) {
    float acc = 0.0f;
    for (int wd = 0; wd < (c.L + 31) / 32; wd++) {
      uint32_t mask = c.jlinks[d * 4 + wd];
      while (mask) {
        const int l = wd * 32 + __ffs((int)mask) - 1;
        mask &= mask - 1;
        acc += wr[l * kWrench + 6];
      }
    }
    c.q[h * c.D + d] = acc;  // grad_q re-uses the q slots
  }
}

// local transforms of point h by one 16-lane row (one sincos per lane)
__device__ __forceinline__ void point_fk_locals(const FusedCtx &c, int h, int lane) {
  float *work = c.work + (size_t)h * c.ws;
  for (int l = lane; l < c.L; l += kFkLanes) {
    const int info = c.link_info[l];
    const int jt = (info & 0xff) - 1;
    const float qv = jt != J_FIXED ? c.q[h * c.D + (info >> 8)] : 0.0f;
    local_transform_colmajor(work + l * 16, c.fixed + l * 12, jt, qv, c.sign[l], c.off_add[l]);
  }
}

// world sphere s of point h overwrites the (dead) local transforms; .w = radius + self-collision padding
__device__ __forceinline__ void point_sphere(const FusedCtx &c, const FusedTrajArgs &a, int b, int h, int s) {
  float4 w4 = transform_sphere(c.cumul + ((size_t)h * c.L + c.sph_link[s]) * 12, c.rs[s]);
  if (a.out_spheres) reinterpret_cast<float4 *>(a.out_spheres)[((size_t)b * c.H + h) * c.S + s] = w4;
  // disabled spheres (negative radius) carry NaN: every pair test against them compares false
  w4.w = (w4.w + c.sph_pad[s]) >= 0.0f ? w4.w + c.sph_pad[s] : __builtin_nanf("");
  reinterpret_cast<float4 *>(c.work + (size_t)h * c.ws)[s] = w4;
}

// Tool-pose goal-set cost of point h (batch row n, horizon position hh of tp.horizon) for every
// tool frame (one per lane): cost -> cost_pt, gradient -> the link wrench as a force at the tool
// link's origin plus the free torque omega = 1/2 E(q)^T g (reference wp_tool_pose.py:456-692,
// kinematics_backward_helper.cuh:102-183, quaternion_util.cuh:86-102).  out_index0 = flat index of
// (n, hh, tool frame 0) in the optional metric outputs.
__device__ __forceinline__ void point_tool_pose(const FusedCtx &c, const ToolPoseArgs &tp, const int16_t *tool_frame_map,
                                                int T, int n, int hh, int h, size_t out_index0, float *out_link_pos,
                                                float *out_link_quat, int lane, int lane64, float &cost_pt, bool &any_grad,
                                                uint32_t *row_bits = nullptr) {
  // (n, hh, h, out_index0 and lane may differ from lane to lane: point_pose_term_pair puts two points on one row; the lanes of
  // the row that contributed a gradient are then reported in row_bits)
  const float *cumul = c.cumul + (size_t)h * c.L * 12;
  float *wr = c.wrench + (size_t)h * c.wl;
  int lane_o = lane;
  asm volatile("" : "+v"(lane_o));  // keeps the per-lane output addresses out of the caller's loop-invariant set
  asm volatile("" : "+v"(lane64));  // ... and values derived from the lane id out of registers held since P0
  for (int t0 = 0; t0 < T; t0 += kFkLanes) {
    const int t = t0 + lane_o;
    f3 gp = make_f3(0.f, 0.f, 0.f), om = gp, pos = gp;
    int l = 0;
    if (t < T) {
      l = tool_frame_map[t];
      const float *C = cumul + l * 12;
      const float4 qx = quat_from_transform(C);
      pos = make_f3(C[3], C[7], C[11]);
      const ToolPoseResult res = tool_pose_distance_point(tp, n, hh, t, pos, make_float4(qx.w, qx.x, qx.y, qx.z));
      cost_pt += res.position_cost + res.rotation_cost;
      gp = res.position_gradient;
      // omega = 0.5 * E(q)^T g  (q xyzw, g wxyz)
      const float dqw = res.quat_rate_wxyz.x, dqx = res.quat_rate_wxyz.y, dqy = res.quat_rate_wxyz.z, dqz = res.quat_rate_wxyz.w;
      om = make_f3(0.5f * (-qx.x * dqw + qx.w * dqx + qx.z * dqy - qx.y * dqz),
                   0.5f * (-qx.y * dqw - qx.z * dqx + qx.w * dqy + qx.x * dqz),
                   0.5f * (-qx.z * dqw + qx.y * dqx - qx.x * dqy + qx.w * dqz));
      const size_t o = out_index0 + t;
      if (tp.out_distance) { tp.out_distance[2 * o] = res.position_cost; tp.out_distance[2 * o + 1] = res.rotation_cost; }
      if (tp.out_position_distance) tp.out_position_distance[o] = res.position_distance;
      if (tp.out_rotation_distance) tp.out_rotation_distance[o] = res.rotation_distance;
      if (tp.out_goalset_idx) tp.out_goalset_idx[o] = res.goalset_idx;
      if (out_link_pos) { float *lp = out_link_pos + o * 3; lp[0] = pos.x; lp[1] = pos.y; lp[2] = pos.z; }
      if (out_link_quat) reinterpret_cast<float4 *>(out_link_quat)[o] = make_float4(qx.w, qx.x, qx.y, qx.z);
    }
    // one contributing lane at a time (lane order): reproducible fp32 sums
    unsigned long long mk = __ballot(gp.x != 0.f || gp.y != 0.f || gp.z != 0.f || om.x != 0.f || om.y != 0.f || om.z != 0.f);
    any_grad = any_grad || ((mk >> (lane64 & 48)) & 0xffffull) != 0ull;
    if (row_bits) *row_bits |= (uint32_t)((mk >> (lane64 & 48)) & 0xffffull);
    while (mk) {
      const int src = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      if (lane64 == src) {
        wrench_add(wr, cumul, l, pos, gp);
        float *w = wr + l * kWrench;
        atomicAdd(w + 3, om.x); atomicAdd(w + 4, om.y); atomicAdd(w + 5, om.z);
      }
    }
  }
}

// c-space STATE constants of this trajectory -> LDS (wp_cspace_state.py:92-160): limits shrunk by
// activation_distance * range, bound / regularisation weights retimed with the trajectory's dt.
// Keeps the ~20 global pointers of the term out of the per-point code (register budget).
__device__ __forceinline__ void stage_cspace_tables(const FusedCtx &c, const CspaceStateArgs &cs, int b) {
  const int D = c.D;
  for (int i = threadIdx.x; i < 10 * D + 11; i += blockDim.x) {
    float v;
    if (i < 10 * D) {
      const int q = i / (2 * D), side = (i / D) & 1, d = i % D;  // quantity 0..4 (pos, vel, acc, jerk, effort), lower / upper
      const float *lim = q == 0 ? cs.p_b : q == 1 ? cs.v_b : q == 2 ? cs.a_b : q == 3 ? cs.j_b : cs.effort_b;
      const float lo = lim[d], hi = lim[D + d], r = hi - lo, eta = cs.activation_distance[q];
      v = side == 0 ? lo + eta * r : hi - eta * r;
    } else {
      const int k = i - 10 * D;
      const float dt = cs.state_dt[b], dt2 = dt * dt, dt3 = dt * dt * dt;
      if (k < 5) {
        v = cs.weight[k];
        if (cs.retime_weights) v = k == 1 ? dt * v : k == 2 ? dt2 * v : k == 3 ? dt3 * v : v;
      } else if (k < 10) {
        const int j = k - 5;
        v = cs.sql2_weights[j];
        if (cs.retime_reg_weights) v = j == 0 ? dt * v : j == 1 ? dt2 * v : j == 2 ? dt3 * v : j == 4 ? dt * v : v;
      } else {
        v = dt;
      }
    }
    c.cstab[i] = v;
  }
}

// c-space STATE cost of point h (wp_cspace_state.py:20-287), one dof per lane, constants from LDS.
// The position gradient is returned per lane (added to grad_q after the wrench gather: it is already
// in joint space); the velocity / acceleration / jerk gradients replace the values in c.dyn.
// Effort terms: tau = nullptr (no dynamics in the launch) contributes nothing; else tau [H][D] are the inverse-dynamics
// torques of the launch and d cost / d tau goes to gtau [H][D].
constexpr int kDofIters = (64 + kFkLanes - 1) / kFkLanes;
__device__ __forceinline__ void point_cspace_state(const FusedCtx &c, const CspaceStateArgs &cs, int b, int h, int lane,
                                                   float &cost_pt, const float *tau = nullptr, float *gtau = nullptr) {
  const int D = c.D, HD = c.H * c.D;
  const float *w = c.cstab + 10 * D;
  // opaque to the optimiser: otherwise the per-lane addresses of the ~12 table / stream slots are
  // hoisted out of the caller's point loop and held in VGPRs across the pose term and the gather
  int d0 = lane;
  asm volatile("" : "+v"(d0));
#pragma unroll 1
  for (int d = d0; d < D; d += kFkLanes) {
    const int e = h * D + d;
    float cc = 0.0f, g0 = 0.0f;
    {
      const float x = c.q[e], lo = c.cstab[d], hi = c.cstab[D + d];
      if (x < lo) squared_l2_term(x - lo, w[0], cc, g0);
      else if (x > hi) squared_l2_term(x - hi, w[0], cc, g0);
      float tw = cs.target_weight[0];
      if (h < c.H - 1) tw *= cs.non_terminal_factor[0];
      if (tw > 0.0f) {
        tw *= cs.target_dof_weight[d];
        const float err = x - cs.target[(size_t)cs.idxs_target[b] * D + d];
        cc += tw * err * err;
        g0 += 2.0f * tw * err;
      }
    }
    const float vel0 = tau != nullptr ? c.dyn[e] : 0.0f;  // (the loop below replaces the values by their gradients)
#pragma unroll
    for (int q = 1; q < 4; q++) {  // velocity, acceleration, jerk: bound + squared-L2 regularisation
      const float x = c.dyn[(q - 1) * HD + e], lo = c.cstab[2 * q * D + d], hi = c.cstab[(2 * q + 1) * D + d];
      float g = 0.0f;
      if (x < lo) squared_l2_term(x - lo, w[q], cc, g);
      else if (x > hi) squared_l2_term(x - hi, w[q], cc, g);
      squared_l2_term(x, w[5 + q - 1], cc, g);
      if (q == 1 && tau != nullptr && w[9] > 0.0f) {  // aggregate_energy_regularization, velocity side
        const float dt = c.cstab[10 * D + 10], en = tau[e] * x * dt;
        g += 2.0f * w[9] * en * tau[e] * dt;
      }
      c.dyn[(q - 1) * HD + e] = g;
    }
    if (tau != nullptr) {  // effort: bound + squared-L2 regularisation + energy (cspace_state_point, x[4])
      const float x = tau[e], lo = c.cstab[8 * D + d], hi = c.cstab[9 * D + d];
      float g = 0.0f;
      if (x < lo) squared_l2_term(x - lo, w[4], cc, g);
      else if (x > hi) squared_l2_term(x - hi, w[4], cc, g);
      squared_l2_term(x, w[8], cc, g);
      if (w[9] > 0.0f) {
        const float dt = c.cstab[10 * D + 10], vel = vel0, en = x * vel * dt;
        cc += w[9] * en * en;
        g += 2.0f * w[9] * en * vel * dt;
      }
      gtau[e] = g;
    }
    cost_pt += cc;
    c.dyn[3 * HD + e] = g0;  // joint-space position gradient, added to grad_q after the wrench gather
    if (cs.out_cost) cs.out_cost[((size_t)b * c.H + h) * D + d] = cc;
  }
}

// LDS views of a workgroup (H = points held by the workgroup)
__device__ __forceinline__ void fused_ctx_carve(FusedCtx &c, float *smem, const FusedLayout &lay, int H, int D, int L, int S,
                                                int P) {
  c.q = smem + lay.q; c.cumul = smem + lay.cumul; c.work = smem + lay.work; c.wrench = smem + lay.wrench;
  c.cost = smem + lay.cost; c.sign = smem + lay.sign; c.sph_rad = smem + lay.sph_rad;
  c.off_add = smem + lay.off_add; c.fixed = smem + lay.fixed; c.sph_pad = smem + lay.sph_pad;
  c.rs = reinterpret_cast<float4 *>(smem + lay.rs);
  c.lbound = reinterpret_cast<int *>(smem + lay.lbound);
  c.parent = reinterpret_cast<int *>(smem + lay.parent);
  c.chain_off = reinterpret_cast<int *>(smem + lay.chain_off);
  c.link_info = reinterpret_cast<int *>(smem + lay.link_info);
  c.chain = reinterpret_cast<int *>(smem + lay.chain);
  c.sph_link = reinterpret_cast<int *>(smem + lay.sph_link);
  c.sub = reinterpret_cast<uint32_t *>(smem + lay.sub);        // [L][4]: links in the subtree of l
  c.jlinks = reinterpret_cast<uint32_t *>(smem + lay.jlinks);  // [D][4]: links driven by joint d
  c.left = reinterpret_cast<float4 *>(smem + lay.left);
  c.lists = reinterpret_cast<uint16_t *>(smem + lay.lists);
  c.key = reinterpret_cast<unsigned long long *>(smem + lay.key);
  c.flag = reinterpret_cast<int *>(smem + lay.flag);
  c.dyn = smem + lay.dyn;
  c.cstab = smem + lay.cstab;
  c.pairs = reinterpret_cast<uint32_t *>(smem + lay.pairs);
  c.lanel = reinterpret_cast<const uint32_t *>(smem + lay.lanel);
  c.g_pairs = nullptr; c.lane_len0 = 0; c.lane_len1 = 0;
  c.recs = reinterpret_cast<ObsRec *>(smem + lay.recs);
  c.H = H; c.D = D; c.L = L; c.S = S; c.P = P; c.ws = lay.ws; c.wl = lay.wl;
}

__device__ __forceinline__ int rotated_tid(int first_wave) {  // jobs start on different waves
  const int nt = blockDim.x, t = (int)threadIdx.x - (first_wave * 64) % nt;
  return t < 0 ? t + nt : t;
}

// Every global read of the robot / scene constants happens here (before the first barrier).  Loads
// are clamped instead of predicated so each loop body is one basic block (all loads issued back to
// back, one wait), and the independent jobs start on different waves so their latencies overlap.
__device__ __forceinline__ void fused_stage_tables(const FusedCtx &c, const FusedTrajArgs &a, const FusedLayout &lay,
                                                   const float4 *rs, int n_rec) {
  const int tid = threadIdx.x, nt = blockDim.x, nwaves = nt >> 6;
  const int H = c.H, D = c.D, L = c.L, S = c.S, P = c.P;
  for (int i = tid; i < L * 4 + D * 4; i += nt) c.sub[i] = 0u;  // sub and jlinks are adjacent
  for (int i = tid; i < H * lay.wl; i += nt) c.wrench[i] = 0.0f;
  for (int i = tid; i < L * 8; i += nt) c.lbound[i] = (i & 4) ? float_key(-3.0e38f) : float_key(3.0e38f);  // empty boxes
  {
    const int C = a.chain_len;
    int n_tab = L * 12;
    n_tab = n_tab > C ? n_tab : C;
    n_tab = n_tab > S ? n_tab : S;
    for (int i = rotated_tid(0); i < n_tab; i += nt) {
      const int il = i < L ? i : L - 1, ic = i < C ? i : C - 1, is = i < S ? i : (S > 0 ? S - 1 : 0);
      const int io = i <= L ? i : L, ix = i < L * 12 ? i : L * 12 - 1;
      const int v_parent = a.link_map[il], v_type = a.joint_map_type[il], v_joint = a.joint_map[il];
      const float v_sign = a.joint_offset[2 * il], v_add = a.joint_offset[2 * il + 1];
      const int v_off = a.link_chain_offsets[io], v_chain = a.link_chain_data[ic];
      const float v_fixed = a.fixed_transform[ix];
      float4 v_rs = make_float4(0.f, 0.f, 0.f, -1.f);
      int v_slink = 0;
      float v_pad = 0.0f;
      if (S > 0) {
        v_rs = rs[is];
        v_slink = a.link_sphere_map[is];
        v_pad = a.sphere_padding ? a.sphere_padding[is] : 0.0f;
      }
      if (i < L) {
        c.parent[i] = v_parent;
        c.link_info[i] = (v_type + 1) | ((v_joint < 0 ? 0 : v_joint) << 8);
        c.sign[i] = v_sign;
        c.off_add[i] = v_add;
      }
      if (i <= L) c.chain_off[i] = v_off;
      if (i < C) c.chain[i] = v_chain;
      if (i < L * 12) c.fixed[i] = v_fixed;
      if (i < S) {
        c.sph_link[i] = v_slink;
        c.sph_rad[i] = v_rs.w;
        c.sph_pad[i] = v_pad;
        c.rs[i] = v_rs;
      }
    }
  }
  // column table of the quad chain of P1 (fk_chain_quad), in the work area (dead until P1 writes the spheres)
  for (int i = rotated_tid(nwaves > 2 ? 2 : 0); i < L * 4; i += nt)
    fk_column_table_entry(reinterpret_cast<float4 *>(c.work), i, a.joint_map_type, a.fixed_transform);
  if (a.use_self && a.lane_lists) {  // lane = sphere form: a straight copy
    const int n = (a.lane_len0 + a.lane_len1) * 64;
    uint32_t *dst = const_cast<uint32_t *>(c.lanel);
    for (int k = tid; k < n; k += nt) dst[k] = a.lane_lists[k];
  } else if (a.use_self) {  // (i, j) -> byte offsets of the float4 spheres; two loads in flight per thread
    const uint32_t *g_pairs = reinterpret_cast<const uint32_t *>(a.pairs);
    for (int k = tid; k < P; k += 2 * nt) {
      const int k1 = k + nt < P ? k + nt : k;
      const uint32_t v0 = g_pairs[k], v1 = g_pairs[k1];
      c.pairs[k] = v0 << 4;
      c.pairs[k1] = v1 << 4;
    }
    for (int k = P + tid; k < ((P + 63) & ~63); k += nt) c.pairs[k] = (uint32_t)(S * 16) | ((uint32_t)(S * 16) << 16);
  }
  if (a.use_scene)
    for (int o = rotated_tid(nwaves - 1); o < n_rec; o += nt)
      c.recs[o] = (o < a.sc.max_cuboids) ? load_rec_global<false>(a.sc, c.env, o)
                                         : load_rec_global<true>(a.sc, c.env, o - a.sc.max_cuboids);
}

// Derived tables (after the first barrier), on the last waves (the first one carries leftover
// points): transposed kinematic tables and a box per link around its collision spheres (link
// frame), both by integer atomics (order independent).
__device__ __forceinline__ void fused_derive_tables(const FusedCtx &c) {
  const int nt = blockDim.x, nwaves = nt >> 6;
  for (int l = rotated_tid(nwaves - 1); l < c.L; l += nt) {
    for (int ci = c.chain_off[l]; ci < c.chain_off[l + 1]; ci++) atomicOr(&c.sub[c.chain[ci] * 4 + (l >> 5)], 1u << (l & 31));
    const int info = c.link_info[l];
    if ((info & 0xff) - 1 >= J_X_PRISM) atomicOr(&c.jlinks[(info >> 8) * 4 + (l >> 5)], 1u << (l & 31));
  }
  for (int sidx = rotated_tid(nwaves > 1 ? nwaves - 2 : 0); sidx < c.S; sidx += nt) {
    const float4 v = c.rs[sidx];
    if (v.w >= 0.0f) {
      int *bx = c.lbound + c.sph_link[sidx] * 8;
      atomicMin(bx + 0, float_key(v.x - v.w)); atomicMin(bx + 1, float_key(v.y - v.w)); atomicMin(bx + 2, float_key(v.z - v.w));
      atomicMax(bx + 4, float_key(v.x + v.w)); atomicMax(bx + 5, float_key(v.y + v.w)); atomicMax(bx + 6, float_key(v.z + v.w));
    }
  }
}

// Workgroups are at most 8 waves when two of them fit in a CU's LDS (<= 80 KB each): 2 x 8 waves =
// 4 per SIMD is what 128 VGPRs allow, and the second workgroup hides the serial phases (table
// loads, the FK chain, barriers) of the first.  (9-wave workgroups do not pair up on a CU even at
// 5 waves/SIMD: measured with tools/probes/lds_occupancy_probe.hip + the profile hook.)
// Points beyond the last full round of 16-lane rows (H = 33 on 32 rows) are "leftover" points:
// instead of a round in which one row works and 31 wait, all threads share them (pairs and spheres
// spread over the workgroup, gradients handed over through LDS, row 0 finishes the VJP).
// Optional terms of a point, each in its own loop over the row's points (the register sets of the
// tool-pose distance, the c-space STATE term and the wrench gather then do not add up):
// tool pose -> cost, wrench and gradient flag; c-space STATE -> cost and stream gradients.
// The argument blocks of the optional terms (~90 scalar registers of pointers) are read WHERE THEY ARE USED, as a burst of scalar
// loads from the kernel-argument segment that the optimiser cannot move: read through the by-value parameter they are loaded at
// kernel entry and stay live across the collision pass, whose own scalars then spill to vector lanes (the TERMS instantiation
// carried 219 spilled scalar registers, ~1100 v_readlane / v_writelane in its straight-line code, and ran collision-only work
// 13 us per 1024 trajectories slower than the collision instantiation).  The struct is the kernel's only parameter: offset 0.
template <class T>
__device__ __forceinline__ T kernarg_block(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) char *kernarg_ptr;
  kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(base));
  return *reinterpret_cast<const __attribute__((address_space(4))) T *>(base + offset);
#else
  return T{};  // (host pass of the translation unit: never called)
#endif
}
// NO_OUT: the launch form of an optimiser iteration writes no per-term outputs (fused_plain_terms)
template <bool NO_OUT>
__device__ __forceinline__ void point_pose_term(const FusedCtx &c, int b, int h, int lane, int lane64) {
  bool any_grad = c.flag[h] != 0;
  float cost2 = 0.0f;
  ToolPoseArgs tp = kernarg_block<ToolPoseArgs>(offsetof(FusedTrajArgs, tp));
  if (NO_OUT) { tp.out_distance = nullptr; tp.out_position_distance = nullptr; tp.out_rotation_distance = nullptr; tp.out_goalset_idx = nullptr; }
  const int16_t *tool_frame_map = kernarg_block<const int16_t *>(offsetof(FusedTrajArgs, tool_frame_map));
  const int n_tool_frames = kernarg_block<int>(offsetof(FusedTrajArgs, n_tool_frames));
  point_tool_pose(c, tp, tool_frame_map, n_tool_frames, b, h, h, ((size_t)b * c.H + h) * n_tool_frames, nullptr, nullptr,
                  lane, lane64, cost2, any_grad);
  cost2 = row16_sum(cost2);
  if (lane == 0) { c.cost[h] += cost2; c.flag[h] = any_grad ? 1 : 0; }
}
template <bool NO_OUT>
__device__ __forceinline__ void point_cspace_term(const FusedCtx &c, int b, int h, int lane, const float *tau, float *gtau) {
  float cost2 = 0.0f;
  CspaceStateArgs cs = kernarg_block<CspaceStateArgs>(offsetof(FusedTrajArgs, cs));
  if (NO_OUT) cs.out_cost = nullptr;
  point_cspace_state(c, cs, b, h, lane, cost2, tau, gtau);
  cost2 = row16_sum(cost2);
  if (lane == 0) c.cost[h] += cost2;
}
// The same two terms for TWO points on one 16-lane row.  With H = rows + 1 points (33 on 32 rows) the loops over the row's points
// run a second time for the one leftover point while 31 rows wait, and these passes are latency (the pose pass: 1.8 us per
// point, quaternion / atan2 / square roots on ONE lane per tool frame; the c-space pass: one lane per dof): the leftover point
// rides on the idle lanes of row 0 instead -- tool frames on lanes T .. 2T-1 (2T <= 16), dofs on lanes 8 .. 8+D-1 (D <= 8).
// Per lane the arithmetic is that of the single-point functions; each point's cost is summed from ITS lanes moved to the
// positions the single-point pass has them in (same reduction tree, same bits); wrenches are added per point in lane order.
template <bool NO_OUT>
__device__ __forceinline__ void point_pose_term_pair(const FusedCtx &c, int b, int h, int h2, int lane, int lane64) {
  ToolPoseArgs tp = kernarg_block<ToolPoseArgs>(offsetof(FusedTrajArgs, tp));
  if (NO_OUT) { tp.out_distance = nullptr; tp.out_position_distance = nullptr; tp.out_rotation_distance = nullptr; tp.out_goalset_idx = nullptr; }
  const int16_t *tool_frame_map = kernarg_block<const int16_t *>(offsetof(FusedTrajArgs, tool_frame_map));
  const int T = kernarg_block<int>(offsetof(FusedTrajArgs, n_tool_frames));
  const bool second = h2 >= 0 && lane >= T && lane < 2 * T;
  const int hp = second ? h2 : h, lt = second ? lane - T : (lane < T ? lane : T);  // (lt = T: the lane has no tool frame)
  const bool had = c.flag[h] != 0, had2 = h2 >= 0 && c.flag[h2] != 0;
  bool any = false;
  uint32_t bits = 0u;
  float cost2 = 0.0f;
  point_tool_pose(c, tp, tool_frame_map, T, b, hp, hp, ((size_t)b * c.H + hp) * T, nullptr, nullptr, lt, lane64, cost2, any, &bits);
  const float moved = __shfl(cost2, (lane + T) & (kFkLanes - 1), kFkLanes);
  const float ca = row16_sum(lane < T ? cost2 : 0.0f), cb = row16_sum(lane < T ? moved : 0.0f);
  const uint32_t low = (1u << T) - 1u;
  if (lane == 0) {
    c.cost[h] += ca;
    c.flag[h] = (had || (bits & low) != 0u) ? 1 : 0;
    if (h2 >= 0) { c.cost[h2] += cb; c.flag[h2] = (had2 || ((bits >> T) & low) != 0u) ? 1 : 0; }
  }
}
template <bool NO_OUT>
__device__ __forceinline__ void point_cspace_term_pair(const FusedCtx &c, int b, int h, int h2, int lane, const float *tau, float *gtau) {
  CspaceStateArgs cs = kernarg_block<CspaceStateArgs>(offsetof(FusedTrajArgs, cs));
  if (NO_OUT) cs.out_cost = nullptr;
  const bool second = lane >= 8;
  const int hp = second ? (h2 >= 0 ? h2 : h) : h, ld = (second && h2 < 0) ? c.D : (lane & 7);  // (ld >= D: nothing to do)
  float cost2 = 0.0f;
  point_cspace_state(c, cs, b, hp, ld, cost2, tau, gtau);
  const float moved = __shfl(cost2, lane ^ 8, kFkLanes);
  const float ca = row16_sum(second ? 0.0f : cost2), cb = row16_sum(second ? 0.0f : moved);
  if (lane == 0) {
    c.cost[h] += ca;
    if (h2 >= 0) c.cost[h2] += cb;
  }
}

// Longest-first dispatch of the trajectory workgroups.  A launch is two rounds of workgroups on the
// chip (1024 trajectories, 2 x 256 resident) and their durations differ by 3x (trajectories deep in
// collision do many more signed-distance evaluations), so the launch ends with a few CUs finishing
// long workgroups that started late.  Workgroups are dispatched in blockIdx order: mapping blockIdx
// through a longest-first permutation starts the long ones in the first round.  The durations of an
// optimiser's candidates change little between iterations, so the previous launch's measurements
// are the estimate: every workgroup records its wall-clock ticks in ticks[phase][b]; one workgroup
// of the launch turns ticks[1 - phase] (complete: written by an earlier launch) into
// order[1 - phase] for the next launch, which the caller runs with the other phase.  A launch reads
// order[phase] and writes order[1 - phase] / ticks[phase], so no array is read and written by the
// same launch, whatever sequence of phases the caller uses; every order[] is a permutation, the
// outputs do not depend on it.  Bucket sort (64 buckets of max/64 ticks) with LDS atomics.
__device__ __forceinline__ void rebuild_dispatch_order(int32_t *ws, int B, int phase, int *lds, int tid, int nt) {
  constexpr int NB = 64;
  const int32_t *ticks = ws + (size_t)(2 + (1 - phase)) * B;
  int32_t *order = ws + (size_t)(1 - phase) * B;
  for (int i = tid; i <= 2 * NB; i += nt) lds[i] = 0;
  __syncthreads();
  int m = 0;
  for (int i = tid; i < B; i += nt) m = max(m, ticks[i]);
  if (m > 0) atomicMax(&lds[2 * NB], m);
  __syncthreads();
  const long long mx = lds[2 * NB];
  for (int i = tid; i < B; i += nt) {
    const int t = max(ticks[i], 0);
    atomicAdd(&lds[NB - 1 - (int)((long long)t * NB / (mx + 1))], 1);  // bucket 0 = longest
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < NB; k++) { lds[NB + k] = acc; acc += lds[k]; }
  }
  __syncthreads();
  for (int i = tid; i < B; i += nt) {
    const int t = max(ticks[i], 0);
    order[atomicAdd(&lds[NB + NB - 1 - (int)((long long)t * NB / (mx + 1))], 1)] = i;
  }
}

#ifdef CUROBO_FUSED_STAMP_TERMS
constexpr bool kStampTerms = true;  // diagnostic builds only: the stamps cost the TERMS variant registers
#else
constexpr bool kStampTerms = false;
#endif

template <class SH> constexpr bool fused_shape_is_plain() { if constexpr (SH::kStatic) return SH::kPlain; else return false; }

// TERMS: the optional tool-pose / c-space STATE terms are compiled in (separate instantiation so the
// collision-only kernel keeps its register budget: with them inlined it spilled 232 B per lane)
template <int DEG, int SWEEP, int KINDS, bool TERMS, class SH = FusedShapeDyn>
__global__ void __launch_bounds__(1024, 4) rollout_trajectory_fused_kernel(const FusedTrajArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.bs.padded_horizon, D = a.bs.dof, L = a.nlinks, S = a.nspheres, P = a.npairs;
  const int n_rec = a.sc.max_cuboids + a.sc.max_voxel_grids;
  const bool use_pose = TERMS && a.use_pose != 0, use_cspace = TERMS && a.use_cspace != 0;
  constexpr bool kPlainTerms = TERMS && fused_shape_is_plain<SH>();
  constexpr int kRings = TERMS ? 1 : 2;  // the TERMS variant has no LDS to spare for the second ring (dense obstacle tests)
  const bool use_lanes = a.lane_lists != nullptr && a.use_self;
  const FusedLayout lay = fused_layout(H, D, L, S, a.chain_len, P, n_rec, use_cspace ? 4 * H * D : 0, (int)blockDim.x >> 6, kRings,
                                       use_lanes ? (a.lane_len0 + a.lane_len1) * 64 : 0);
  const int tid = threadIdx.x;
  const int wave_idx = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockDim.x;
  if constexpr (SH::kStatic) {
    // a compile-time shape (fused_shapes.hpp): the host launches this instantiation only when every one of these holds
    __builtin_assume(H == SH::kH); __builtin_assume(a.bs.n_knots == SH::kNK); __builtin_assume(D == SH::kD); __builtin_assume(L == SH::kL);
    __builtin_assume(S == SH::kS); __builtin_assume(P == SH::kP); __builtin_assume(a.chain_len == SH::kC);
    __builtin_assume(a.lane_len0 == SH::kLen0); __builtin_assume(a.lane_len1 == SH::kLen1); __builtin_assume(nt == SH::kNT);
    if constexpr (SH::kNCub >= 0) {
      __builtin_assume(a.sc.max_cuboids == SH::kNCub); __builtin_assume(a.sc.max_voxel_grids == SH::kNVox);
      __builtin_assume(n_rec == SH::kNCub + SH::kNVox);
    }
    __builtin_assume(a.lane_lists != nullptr);
    if constexpr (SH::kPlain) {  // the launch form of an optimiser iteration (fused_plain_launch)
      __builtin_assume(a.use_self == 1); __builtin_assume(a.use_scene == 1); __builtin_assume(a.enable_speed_metric == 1);
      __builtin_assume(a.out_position == nullptr); __builtin_assume(a.out_spheres == nullptr); __builtin_assume(a.prof == nullptr);
      __builtin_assume(a.use_multi_env == 0); __builtin_assume(a.num_envs == 1); __builtin_assume(a.scene_rows == 0);
      __builtin_assume(a.dispatch_ws != nullptr); __builtin_assume(a.sphere_padding != nullptr);
      if constexpr (TERMS) {  // ... of a trajectory-optimisation iteration (fused_plain_terms): tool pose + c-space STATE, no torque limits
        __builtin_assume(a.use_pose == 1); __builtin_assume(a.use_cspace == 1); __builtin_assume(a.use_torque == 0);
      }
    }
  }
  // trajectory of this workgroup: blockIdx.x itself, or the entry of the longest-first order that the
  // previous launches built from the measured workgroup durations (same results, shorter tail)
  const bool reorder = a.dispatch_ws != nullptr;
  const int b = reorder ? a.dispatch_ws[(size_t)a.dispatch_phase * a.batch + blockIdx.x] : (int)blockIdx.x;
  const long long t_begin = reorder ? wall_clock64() : 0ll;
  FusedCtx c;
  fused_ctx_carve(c, smem, lay, H, D, L, S, P);
  if (use_lanes) { c.g_pairs = reinterpret_cast<const uint32_t *>(a.pairs); c.lane_len0 = a.lane_len0; c.lane_len1 = a.lane_len1; }
  c.env = a.use_multi_env ? a.env_query_idx[b] : 0;
#ifdef CUROBO_FUSED_WAVE_STAMPS
#define CUROBO_STAMP(i) do { if ((i) < 8 && (!TERMS || kStampTerms) && a.prof && tid == 0) a.prof[(size_t)b * 16 + (i)] = wall_clock64(); } while (0)
#else
#define CUROBO_STAMP(i) do { if ((!TERMS || kStampTerms) && a.prof && tid == 0) a.prof[(size_t)b * 16 + (i)] = wall_clock64(); } while (0)
#endif
  CUROBO_STAMP(0);
  const int sph_env = a.num_envs > 1 ? a.env_query_idx[b] : 0;
  const float4 *rs = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)sph_env * S;
  // launch-wide scalars: loaded through the vector path, moved to SGPRs (else each occupies a VGPR
  // for the whole kernel)
  c.w_self = uniform_f(a.use_self ? a.w_self[0] : 0.0f);
  c.w_scene = uniform_f(a.use_scene ? a.w_scene[0] : 0.0f);
  c.eta = uniform_f(a.use_scene ? a.eta[0] : 0.0f);
  c.speed_metric = a.enable_speed_metric != 0;
  c.speed_dt = uniform_f(c.speed_metric ? a.speed_dt[0] : 0.0f);

  // ---------------- P0: tables + B-spline samples
  const int nwaves = nt >> 6;
  // (the trajectory's state indices, dt and goal mode: requested here, ahead of the table loads, used by the samples)
  const int bs_bo = a.bs.start_idx[b], bs_go = a.bs.goal_idx[b];
  const float bs_dt = a.bs.traj_dt[bs_go];
  const bool bs_implicit = a.bs.use_implicit_goal[bs_go] != 0;
  fused_stage_tables(c, a, lay, rs, n_rec);
  if (use_cspace) stage_cspace_tables(c, kernarg_block<CspaceStateArgs>(offsetof(FusedTrajArgs, cs)), b);
  for (int e = rotated_tid(nwaves / 2); e < H * D; e += nt) {
    const int h = e / D, d = e - h * D;
    float o4[4];
    bspline_sample_pre<DEG>(a.bs, b, h, d, a.bs.padded_horizon, bs_dt, bs_bo, bs_go, bs_implicit, o4);
    c.q[e] = o4[0];
    if (use_cspace) { c.dyn[e] = o4[1]; c.dyn[H * D + e] = o4[2]; c.dyn[2 * H * D + e] = o4[3]; }
    if (a.out_position) a.out_position[(size_t)b * H * D + e] = o4[0];
  }
  __syncthreads();
  CUROBO_STAMP(1);

  const int grp = tid / kFkLanes, lane = tid % kFkLanes, ngroups = nt / kFkLanes;
  const int lane64 = tid & 63;
  // leftover points are shared by the workgroup when they are few (else: one more ordinary round)
  int n_left = H % ngroups;
  if (n_left * 4 > ngroups || H < ngroups) n_left = 0;
  const int H_main = H - n_left;

  // ---------------- P1: FK.  (1) sin / cos of every (point, jointed link) on all lanes, parked in the first
  // two floats of that link's (not yet written) cumulative slot; (2) the chain of every point on ONE QUAD
  // (fk_chain_quad: ~30 instructions per link on the critical path; 4 H lanes, the other wavefronts derive the
  // transposed tables meanwhile); (3) the world spheres of every (point, sphere) on all lanes.  Every point is
  // treated alike here; the row / leftover split only concerns the cost passes of P2.
  for (int e = tid; e < H * L; e += nt) {
    const int h = e / L, l = e - h * L;
    const int info = c.link_info[l];
    const int jt = (info & 0xff) - 1;
    if (jt != J_FIXED) {
      float sn, cs;
      joint_sincos(jt, c.q[h * D + (info >> 8)], c.sign[l], c.off_add[l], &sn, &cs);
      *reinterpret_cast<float2 *>(c.cumul + ((size_t)h * L + l) * 12) = make_float2(sn, cs);
    }
  }
  CUROBO_STAMP(8);
  __syncthreads();
  // the chains are the serial part the whole workgroup waits for, on wavefronts that share their SIMDs with the
  // co-resident workgroup's cost passes: they run at raised issue priority
  __builtin_amdgcn_s_setprio(3);
  for (int pt = tid >> 2; pt < H; pt += nt >> 2) {
    float *cm = c.cumul + (size_t)pt * L * 12;
    fk_chain_quad(cm, reinterpret_cast<const float4 *>(c.work), c.parent, 1, L, tid & 3, cm, 12);
  }
  __builtin_amdgcn_s_setprio(0);
  fused_derive_tables(c);  // (on the last wavefronts: next to the chains, not after them)
  CUROBO_STAMP(9);
  __syncthreads();
  {
    const int step_h = nt / S, step_s = nt % S;
    int h = tid / S, sp = tid - (tid / S) * S;
    for (int e = tid; e < H * S; e += nt) {
      point_sphere(c, a, b, h, sp);
      sp += step_s; h += step_h;
      if (sp >= S) { sp -= S; h++; }
    }
    for (int hh = tid; hh < H; hh += nt)
      reinterpret_cast<float4 *>(c.work + (size_t)hh * c.ws)[S] = make_float4(0.f, 0.f, 0.f, __builtin_nanf(""));
  }
  CUROBO_STAMP(10);
  if (tid == 0) { c.key[0] = 0ull; c.key[1] = 0ull; }  // leftover point: arg-max key of its pair list, ticket for its scene pass
  __syncthreads();
  CUROBO_STAMP(2);

  // ---------------- P2: costs + VJP per point.  The waves stay converged over the rounds (rows
  // without a point in the last round are masked), because the scene pass is a wave-level job.
  // Row r of wave w takes point w + r * nwaves of the round (not 4 * w + r): points deep in collision
  // come in runs along the trajectory, and a wave is as slow as the sum of its rows' scene work.
  const int row_stride = ngroups >> 2;
  // A single leftover point (H = 33 / 65 on 32 / 64 rows) is folded into the main round instead of a
  // barrier - all threads - barrier - fold section: its pair list is sliced over the rows, its scene cost is
  // evaluated by wave 0 (one sphere per lane) with the link-mask culling of an ordinary point, nothing is handed
  // over through LDS.
  const bool fold_left = n_left == 1;
  for (int h0 = 0; h0 < H_main; h0 += ngroups) {
    const int h = h0 + (grp & 3) * row_stride + (grp >> 2);
    const bool valid = h < H_main;
    const float4 *sph = c.spheres(valid ? h : 0);
    float cost_pt = 0.0f;
    bool any_grad = false;  // uniform over the 16-lane row
    if (use_lanes) {
      // Lane = sphere (reference self_collision_kernel.cuh:19-111, same arg-max).  The wavefront takes its four points
      // together: a lane keeps its OWN sphere of each point in registers and walks the partners that
      // curobo_hip_self_lane_lists_host dealt to it -- every pair sits in the list of exactly one of its two spheres,
      // the lists are balanced (818 Franka pairs: 13 or 14 per lane) -- so a pair costs one ds_read_b128 instead of
      // two plus its index: this pass was bound by LDS bandwidth.  Only the maximum is tracked; the pair index is
      // looked for afterwards, and only by points that are in self collision.
      const int wv = grp >> 2;
      const char *sp[4];
      bool ok[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int hr = h0 + r * row_stride + wv;
        ok[r] = hr < H_main;
        sp[r] = reinterpret_cast<const char *>(c.spheres(ok[r] ? hr : 0));
      }
      float bm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      const uint32_t *ll = c.lanel + lane64;
      for (int pass = 0; pass < 2; pass++) {
        const int len = pass == 0 ? c.lane_len0 : c.lane_len1;
        if (len == 0) continue;
        const int own_i = pass * 64 + lane64;
        const int own_off = (own_i < S ? own_i : S) * 16;
        float4 own[4];
#pragma unroll
        for (int r = 0; r < 4; r++) own[r] = *reinterpret_cast<const float4 *>(sp[r] + own_off);
        for (int e = 0; e < len; e++) {
          const uint32_t off = ll[e * 64] & 0xffffu;
#pragma unroll
          for (int r = 0; r < 4; r++)
            bm[r] = max_quiet(bm[r], sphere_pair_penetration(own[r], *reinterpret_cast<const float4 *>(sp[r] + off)));
        }
        ll += len * 64;
      }
      float m_row = 0.0f;
      int k_row = 0x7fffffff;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float m = wave64_max(bm[r]);
        if (m > 0.0f && ok[r]) {  // (wave-uniform) in self collision: lowest pair index of the largest penetration
          int kb = 0x7fffffff;
          const uint32_t *l2 = c.lanel + lane64;
          for (int pass = 0; pass < 2; pass++) {
            const int len = pass == 0 ? c.lane_len0 : c.lane_len1;
            const int own_i = pass * 64 + lane64;
            const float4 own1 = *reinterpret_cast<const float4 *>(sp[r] + (own_i < S ? own_i : S) * 16);
            for (int e = 0; e < len; e++) {
              const uint32_t ent = l2[e * 64];
              const float f = sphere_pair_penetration(own1, *reinterpret_cast<const float4 *>(sp[r] + (ent & 0xffffu)));
              if (f == m) kb = min(kb, (int)(ent >> 16));
            }
            l2 += len * 64;
          }
          kb = wave64_min(kb);
          if ((grp & 3) == r) { m_row = m; k_row = kb; }
        }
      }
      if (valid && k_row != 0x7fffffff && m_row > 0.0f) {
        any_grad = true;
        if (lane == 0) cost_pt += self_pair_apply(c, h, m_row, k_row);
      }
    } else if (a.use_self && valid) {  // reference self_collision_kernel.cuh:19-111
      // One pass over the padded pair list (no bounds checks; disabled / padding spheres are NaN and
      // lose every max).  Per pair only a v_max; the arg-max is tracked per group of U pairs (one
      // compare per group) and resolved inside the winning group afterwards.
      constexpr int U = 4;
      const int P_pad = (P + 63) & ~63;
      float best = 0.0f;
      int best_k0 = 0x7fffffff;
      for (int k0 = lane; k0 < P_pad; k0 += kFkLanes * U) {
        uint32_t ij[U];
#pragma unroll
        for (int u = 0; u < U; u++) ij[u] = c.pairs[k0 + u * kFkLanes];
        float gmax = staged_pair_penetration(sph, ij[0]);
#pragma unroll
        for (int u = 1; u < U; u++) gmax = fmaxf(gmax, staged_pair_penetration(sph, ij[u]));
        if (gmax > best) { best = gmax; best_k0 = k0; }
      }
      float m = row16_max(best);
      int kmin = 0x7fffffff;
      if (m > 0.0f) {  // rows in self collision: lowest pair index of the largest penetration
        int best_k = 0x7fffffff;
        if (best == m) {
          float f_best = 0.0f;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const float f = staged_pair_penetration(sph, c.pairs[best_k0 + u * kFkLanes]);
            if (f > f_best) { f_best = f; best_k = best_k0 + u * kFkLanes; }
          }
          best = f_best;
        } else {
          best = 0.0f;
        }
        m = row16_max(best);
        kmin = row16_min((best == m && best > 0.0f) ? best_k : 0x7fffffff);
      }
      if (kmin != 0x7fffffff && m > 0.0f) {
        any_grad = true;
        if (lane == 0) cost_pt += self_pair_apply(c, h, m, kmin);
      }
    }
    if (fold_left && a.use_self && h0 == 0) {
      // this row's slice of the ONE leftover point's pair list; the arg-max meets in c.key (integer max of
      // (penetration, lowest pair index): order independent) and is applied by row 0 after the barrier
      const float4 *sphl = c.spheres(H_main);
      float bl = 0.0f;
      int kl = 0x7fffffff;
      if (use_lanes) {  // the list entries of a lane are dealt over the wavefronts
        const char *spl = reinterpret_cast<const char *>(sphl);
        const uint32_t *ll = c.lanel + lane64;
        for (int pass = 0; pass < 2; pass++) {
          const int len = pass == 0 ? c.lane_len0 : c.lane_len1;
          const int own_i = pass * 64 + lane64;
          const float4 own1 = *reinterpret_cast<const float4 *>(spl + (own_i < S ? own_i : S) * 16);
          for (int e = (grp >> 2); e < len; e += nwaves) {
            const uint32_t ent = ll[e * 64];
            const float f = sphere_pair_penetration(own1, *reinterpret_cast<const float4 *>(spl + (ent & 0xffffu)));
            const int k = (int)(ent >> 16);
            if (f > bl || (f == bl && f > 0.0f && k < kl)) { bl = f; kl = k; }
          }
          ll += len * 64;
        }
      } else {
        for (int k = grp * kFkLanes + lane; k < P; k += ngroups * kFkLanes) {
          const float f = staged_pair_penetration(sphl, c.pairs[k]);
          if (f > bl) { bl = f; kl = k; }
        }
      }
      if (bl > 0.0f) atomicMax(c.key, pair_key(bl, kl));
    }
    const bool stamp_pt = (!TERMS || kStampTerms) && a.prof && lane == 0 && h == (b % H);
    if (stamp_pt) a.prof[(size_t)b * 16 + 5] = wall_clock64();
    if (valid && lane == 0) { c.cost[h] = cost_pt; c.flag[h] = any_grad ? 1 : 0; }
    if (a.use_scene) {
      if (valid) point_link_masks<SWEEP, KINDS>(c, a.sc, h, lane);
      if (S <= kScenePassMaxSpheres && n_rec <= kScenePassMaxRecords && a.scene_rows != 1) {
        // dense packing of the obstacle tests: cuboid-only worlds (with an ESDF grid that covers the workspace every
        // sphere is active anyway; the voxel x sweep instantiation also did not reproduce the row pass with it)
        wave_scene_pass<SWEEP, KINDS, !TERMS && KINDS == 1>(c, a.sc, h, valid, lane, lane64, c.lists + (tid >> 6) * kRings * kSceneListEntries, row_stride,
                                              a.scene_rows == 2 ? 1 : 64);
      } else if (valid) {  // beyond the ring's entry format: one sphere per lane, all its obstacles
        const float *wr = c.wrench + (size_t)h * c.wl;
        float cost_scene = 0.0f;
        bool any_scene = false;
        for (int s0 = 0; s0 < S; s0 += kFkLanes) {
          const int s = s0 + lane;
          float d = 0.0f;
          f3 g = make_f3(0.f, 0.f, 0.f);
          float4 c4 = make_float4(0.f, 0.f, 0.f, -1.f);
          const uint32_t mask = s < S ? __float_as_uint(wr[c.sph_link[s] * kWrench + 6]) : 0u;
          // records beyond the 32 mask bits are never culled: the sphere is evaluated whatever its link mask says
          if (s < S && (mask != 0u || n_rec > 32)) c4 = scene_sphere<SWEEP, KINDS>(c, a.sc, h, s, d, g, mask);
          cost_scene += d;
          any_scene = wrench_add_serialised(c, h, s, make_f3(c4.x, c4.y, c4.z), g, lane64) || any_scene;
        }
        cost_scene = row16_sum(cost_scene);
        if (lane == 0) { c.cost[h] += cost_scene; if (any_scene) c.flag[h] = 1; }
      }
    }
    if (stamp_pt) a.prof[(size_t)b * 16 + 6] = wall_clock64();
  }
#ifdef CUROBO_FUSED_WAVE_STAMPS  // diagnostic build: when each wavefront leaves the main round (slots 8..15 of the profile row)
  if (a.prof && !TERMS && lane64 == 0) a.prof[(size_t)b * 16 + 8 + (tid >> 6)] = wall_clock64();
#endif
  if (fold_left) {
    CUROBO_STAMP(7);
    const int h = H_main;
    // The wavefront that leaves the main round FIRST takes this point (a ticket in LDS), all 64 lanes on it, one
    // sphere per lane.  The rest of the workgroup waits at the barrier below for the slowest wavefront of the main
    // round anyway; with a fixed wavefront (it used to be wave 0) that wavefront's rows + this section were the
    // critical path whenever wave 0 was not among the early ones.  Which wavefront runs it does not change a bit
    // of the result: it reads and writes this point's slots only.
    int ticket = 1;
    if (a.use_scene) {
      if (lane64 == 0) ticket = atomicAdd(reinterpret_cast<int *>(c.key) + 2, 1);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
    }
    if (a.use_scene && ticket == 0) {
      // the wave's first row computes the link masks
      if ((grp & 3) == 0) point_link_masks<SWEEP, KINDS>(c, a.sc, h, lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float *wr = c.wrench + (size_t)h * c.wl;
      float cost_scene = 0.0f;
      bool any_scene = false;
      for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane64;
        float d = 0.0f;
        f3 g = make_f3(0.f, 0.f, 0.f);
        float4 c4 = make_float4(0.f, 0.f, 0.f, -1.f);
        const uint32_t mask = s < S ? __float_as_uint(wr[c.sph_link[s] * kWrench + 6]) : 0u;
        if (s < S && (mask != 0u || n_rec > 32)) c4 = scene_sphere<SWEEP, KINDS>(c, a.sc, h, s, d, g, mask);
        cost_scene += d;
        any_scene = __ballot(g.x != 0.0f || g.y != 0.0f || g.z != 0.0f) != 0ull || any_scene;
        // (one turn per contributing lane, ascending sphere index: the order a single row would have used)
        wrench_add_serialised(c, h, s, make_f3(c4.x, c4.y, c4.z), g, lane64);
      }
      // sum in the order of a 16-lane row striding over the spheres (lane l: s = l, 16 + l, ...), then across the row
      float x = __shfl(cost_scene, lane, 64);
      x += __shfl(cost_scene, lane + 16, 64);
      x += __shfl(cost_scene, lane + 32, 64);
      x += __shfl(cost_scene, lane + 48, 64);
      cost_scene = row16_sum(x);
      if (lane64 == 0) { c.cost[h] = cost_scene; c.flag[h] = any_scene ? 1 : 0; }
    }
    if (!a.use_scene && tid == 0) { c.cost[h] = 0.0f; c.flag[h] = 0; }
  }
  for (int h = H_main; h < H && !fold_left; h++) {  // several leftover points: all threads on one point at a time
    if (tid == 0) *c.key = 0ull;
    __syncthreads();
    if (h == H_main) CUROBO_STAMP(7);
    if (a.use_self) {
      const float4 *sph = c.spheres(h);
      float best = 0.0f;
      int best_k = 0x7fffffff;
      for (int k = tid; k < P; k += nt) {
        const float f = staged_pair_penetration(sph, c.g_pairs ? (c.g_pairs[k] << 4) : c.pairs[k]);
        if (f > best) { best = f; best_k = k; }
      }
      if (best > 0.0f) atomicMax(c.key, pair_key(best, best_k));  // integer max: order independent
    }
    if (a.use_scene)
      for (int s = tid; s < S; s += nt) {
        float d;
        f3 g;
        scene_sphere<SWEEP, KINDS>(c, a.sc, h, s, d, g);
        c.left[s] = make_float4(g.x, g.y, g.z, d);
      }
    __syncthreads();
    if (grp == 0) {  // row 0 folds the handed-over results in the same order as an ordinary point
      float cost_pt = 0.0f;
      bool any_grad = false;
      const unsigned long long key = *c.key;
      if (key != 0ull) {
        any_grad = true;
        if (lane == 0) cost_pt += self_pair_apply(c, h, __uint_as_float((uint32_t)(key >> 32)), (int)(0x7fffffffu - (uint32_t)key));
      }
      if (a.use_scene) {
        const float4 *sph = c.spheres(h);
        for (int s0 = 0; s0 < S; s0 += kFkLanes) {
          const int s = s0 + lane;
          float4 gd = make_float4(0.f, 0.f, 0.f, 0.f), c4 = gd;
          if (s < S) { gd = c.left[s]; c4 = sph[s]; }
          cost_pt += gd.w;
          any_grad = wrench_add_serialised(c, h, s, make_f3(c4.x, c4.y, c4.z), make_f3(gd.x, gd.y, gd.z), lane64) || any_grad;
        }
      }
      cost_pt = row16_sum(cost_pt);
      if (lane == 0) { c.cost[h] = cost_pt; c.flag[h] = any_grad ? 1 : 0; }
    }
  }
  __syncthreads();  // the passes below take point h on row h % rows: another wave than the cost pass above
  if (fold_left && a.use_self && grp == 0) {
    const unsigned long long key = *c.key;
    if (key != 0ull && lane == 0) {
      c.cost[H_main] += self_pair_apply(c, H_main, __uint_as_float((uint32_t)(key >> 32)), (int)(0x7fffffffu - (uint32_t)key));
      c.flag[H_main] = 1;
    }
  }
  // ---- inverse dynamics of every point (torque limits), one lane per point: see fused_torque_fits for where its
  // state lives.  Forward sweeps here (the c-space pass below needs tau), the VJP after the wrench gather.
  const bool use_torque = TERMS && use_cspace && a.use_torque != 0;
  float *tq_f = smem + lay.lists;
  int *tq_i = reinterpret_cast<int *>(tq_f + L * kLinkFloats);
  float *tq_q = smem + lay.pairs, *tq_tau = reinterpret_cast<float *>(c.left), *tq_gtau = smem + lay.sph_link;
  RneaArgs rn{};
  if (use_torque) {
    __syncthreads();  // row 0 has applied the leftover point's self-collision pair: pair list, spheres and rings are dead
#define CUROBO_KA(T, f) kernarg_block<T>(offsetof(FusedTrajArgs, f))
    rn.fixed_transforms = CUROBO_KA(const float *, fixed_transform); rn.link_masses_com = CUROBO_KA(const float *, link_masses_com);
    rn.link_inertias = CUROBO_KA(const float *, link_inertias); rn.joint_map_type = CUROBO_KA(const int8_t *, joint_map_type);
    rn.joint_map = CUROBO_KA(const int16_t *, joint_map); rn.link_map = CUROBO_KA(const int16_t *, link_map);
    rn.joint_offset_map = CUROBO_KA(const float *, joint_offset); rn.gravity = CUROBO_KA(const float *, gravity);
    rn.level_links = CUROBO_KA(const int16_t *, level_links);
#undef CUROBO_KA
    rn.num_links = L; rn.num_dof = D; rn.batch = H;
    rn.q = tq_q; rn.qd = tq_q + H * D; rn.qdd = tq_q + 2 * H * D; rn.tau = tq_tau; rn.cache = c.work;
    for (int e = tid; e < H * D; e += nt) {
      tq_q[e] = c.q[e];
      tq_q[H * D + e] = c.dyn[e];
      tq_q[2 * H * D + e] = c.dyn[H * D + e];
    }
    stage_links(rn, tq_f, tq_i);  // (ends with a workgroup barrier)
    if (tid < H) rnea_forward_element_io<false>(rn, RneaLdsIO(rn, (size_t)tid), tq_f, tq_i, tq_i + L * 3, (size_t)tid, (size_t)H);
    __syncthreads();
  }
  CUROBO_STAMP(15);
  // further passes over all points, leftover ones included (their own loops so that the register
  // allocation of the collision pass above is not shared with the optional terms, and so that those
  // are instantiated once): tool pose, c-space STATE, then the VJP gather
  CUROBO_STAMP(12);
  const bool pair_left = H == ngroups + 1;  // one leftover point: it shares row 0 with point 0 in the two passes below
  if (TERMS && use_pose) {
    if (pair_left && 2 * a.n_tool_frames <= kFkLanes) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      point_pose_term_pair<kPlainTerms>(c, b, grp, grp == 0 ? H - 1 : -1, lane, lane64);
    } else {
      for (int h = grp; h < H; h += ngroups) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        point_pose_term<kPlainTerms>(c, b, h, lane, lane64);
      }
    }
  }
  CUROBO_STAMP(13);
  if (TERMS && use_cspace) {
    if (pair_left && D <= 8)
      point_cspace_term_pair<kPlainTerms>(c, b, grp, grp == 0 ? H - 1 : -1, lane, use_torque ? tq_tau : nullptr, tq_gtau);
    else
      for (int h = grp; h < H; h += ngroups) point_cspace_term<kPlainTerms>(c, b, h, lane, use_torque ? tq_tau : nullptr, tq_gtau);
  }
  CUROBO_STAMP(14);
  for (int h = grp; h < H; h += ngroups) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    point_vjp_gather(c, h, c.flag[h] != 0, lane);
    // joint-space part of the c-space gradient: added after the gather (same lane wrote the slot)
    if (TERMS && use_cspace)
      for (int d = lane; d < D; d += kFkLanes) c.q[h * D + d] += c.dyn[3 * H * D + h * D + d];
  }
  __syncthreads();
  if (use_torque) {  // VJP of the inverse dynamics: d cost / d tau -> added to the joint-space gradient streams
    rn.grad_q = c.q; rn.grad_qd = c.dyn; rn.grad_qdd = c.dyn + H * D; rn.grad_tau = tq_gtau;
    rn.ws_fbar = c.cumul; rn.ws_abar = c.cumul + (size_t)L * 6 * H; rn.ws_vbar = c.wrench;
    if (tid < H) rnea_backward_element_io<false, true>(rn, RneaLdsIO(rn, (size_t)tid), tq_f, tq_i, tq_i + L * 3, (size_t)tid, (size_t)H);
    __syncthreads();
  }
  CUROBO_STAMP(3);

  // ---------------- P3: B-spline VJP + trajectory cost
  // (its pointers come from the argument segment here, kernarg_block: read through `a` they are loaded at entry and stay in
  // scalar registers across the collision pass)
  const int nk = a.bs.n_knots;
  constexpr size_t kBs = offsetof(FusedTrajArgs, bs);
  const int go = kernarg_block<const int32_t *>(kBs + offsetof(BsFwdArgs, goal_idx))[b];
  const float traj_dt = kernarg_block<const float *>(kBs + offsetof(BsFwdArgs, traj_dt))[go];
  const bool use_goal = kernarg_block<const uint8_t *>(kBs + offsetof(BsFwdArgs, use_implicit_goal))[go] != 0;
  float *const out_grad_knots = kernarg_block<float *>(offsetof(FusedTrajArgs, out_grad_knots));
  float *const out_cost = kernarg_block<float *>(offsetof(FusedTrajArgs, out_cost));
  const float *gin[4] = {c.q, use_cspace ? c.dyn : nullptr, use_cspace ? c.dyn + H * D : nullptr,
                         use_cspace ? c.dyn + 2 * H * D : nullptr};
  // TERMS variant: the thread id is recomputed here (uniform wave index * 64 + mbcnt, opaque to CSE)
  // instead of being carried in a VGPR from the top: it was the one value that variant spilled
  int tid3 = tid;
  if (TERMS) {
    int l64;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l64));
    tid3 = wave_idx * 64 + l64;
  }
  for (int e = tid3; e < nk * D; e += nt) {
    const int k = e / D, d = e - k * D;
    out_grad_knots[(size_t)b * nk * D + e] = bspline_knot_grad<DEG>(gin, (size_t)d, D, k, nk, H, traj_dt, use_goal);
  }
  if (tid3 == 0) {
    float acc = 0.0f;
    for (int h = 0; h < H; h++) acc += c.cost[h];
    out_cost[b] = acc;
  }
  CUROBO_STAMP(4);
#undef CUROBO_STAMP
  if (reorder) {
    int32_t *const ws = kernarg_block<int32_t *>(offsetof(FusedTrajArgs, dispatch_ws));
    const int phase = kernarg_block<int>(offsetof(FusedTrajArgs, dispatch_phase)), n_traj = kernarg_block<int>(offsetof(FusedTrajArgs, batch));
    if (tid3 == 0) ws[(size_t)(2 + phase) * n_traj + b] = (int)(wall_clock64() - t_begin);
    if (blockIdx.x == (gridDim.x - 1) / 2) {
      __syncthreads();
      rebuild_dispatch_order(ws, n_traj, phase, reinterpret_cast<int *>(smem), tid3, nt);
    }
  }
}


// ------------------------------------------------------------------------------------------
// Horizon-1 ("teleport") rollout of the IK solver in one launch: q -> FK -> tool-pose goal-set
// cost + c-space bound cost + self + scene collision -> cost[b] and d cost / d q[b].  Replaces
// the seven launches of curobo_amd/rollout/ik_rollout.py (reference RobotRollout with
// StateFromPositionTeleport and the cost set of content/configs/task/ik/lbfgs_ik.yml); same
// device functions, intermediates in LDS.  16 configurations per 256-thread workgroup (one
// 16-lane row each) share the staged robot / scene tables.  The tool-pose gradient enters the
// link-wrench VJP as a force at the tool link's origin plus the free torque omega = 1/2 E(q)^T g
// (reference kinematics_backward_helper.cuh:102-183, quaternion_util.cuh:86-102).
struct FusedIkArgs {
  FusedTrajArgs r;  // robot / self / scene members, out_cost, out_position (= nothing), out_spheres
  const float *x;   // [n_points, dof]
  float *out_grad_q;
  ToolPoseArgs tp;  // current_position / current_quat unused (computed here); out_* optional
  const float *p_b, *cs_weight, *cs_eta;  // c-space bound term: limits [2, dof], weight[>=1], activation[>=1]
  float *out_cspace_cost;                 // optional [n_points, dof]
  const int16_t *tool_frame_map;
  float *out_link_pos, *out_link_quat;  // optional [n_points, T, 3|4]
  int n_tool_frames, n_points;
};

constexpr int kIkPoints = 16;

template <int KINDS>
__global__ void __launch_bounds__(256, 4) rollout_ik_fused_kernel(const FusedIkArgs ia) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FusedTrajArgs &a = ia.r;
  const int H = kIkPoints, D = a.bs.dof, L = a.nlinks, S = a.nspheres, P = a.npairs, T = ia.n_tool_frames;
  const int n_rec = a.sc.max_cuboids + a.sc.max_voxel_grids;
  const FusedLayout lay = fused_layout(H, D, L, S, a.chain_len, P, n_rec, 0, 0, 1, 0, false);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int pt0 = blockIdx.x * kIkPoints;
  const int npts = min(kIkPoints, ia.n_points - pt0);
  FusedCtx c;
  fused_ctx_carve(c, smem, lay, H, D, L, S, P);
  // the 16 configurations of a workgroup share ONE environment (scene and sphere set): that of its first configuration.
  // The caller checks that env_query_idx is constant over aligned runs of 16 (seeds of one problem: IkRollout).
  const int wg_env = (a.use_multi_env || a.num_envs > 1) ? a.env_query_idx[pt0] : 0;
  c.env = a.use_multi_env ? wg_env : 0;
  c.w_self = a.use_self ? a.w_self[0] : 0.0f;
  c.w_scene = a.use_scene ? a.w_scene[0] : 0.0f;
  c.eta = a.use_scene ? a.eta[0] : 0.0f;
  c.speed_metric = false;
  c.speed_dt = 0.0f;
  fused_stage_tables(c, a, lay, reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)(a.num_envs > 1 ? wg_env : 0) * S, n_rec);
  for (int e = rotated_tid((nt >> 6) / 2); e < npts * D; e += nt) c.q[e] = ia.x[(size_t)pt0 * D + e];
  __syncthreads();
  fused_derive_tables(c);

  const int h = tid / kFkLanes, lane = tid % kFkLanes, lane64 = tid & 63;
  const bool live = h < npts;
  const int n = pt0 + h;
  // ---- FK
  if (live) {
    point_fk_locals(c, h, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float *const cm[1] = {c.cumul + (size_t)h * L * 12};
    const float *const lc[1] = {c.work + (size_t)h * c.ws};
    fk_chain_16_multi<1>(cm, lc, c.parent, c.fixed, L, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < S; s += kFkLanes) point_sphere(c, a, blockIdx.x, h, s);
    if (lane == 0) reinterpret_cast<float4 *>(c.work + (size_t)h * c.ws)[S] = make_float4(0.f, 0.f, 0.f, __builtin_nanf(""));
  }
  __syncthreads();  // derived tables (other waves) + this row's spheres
  if (!live) return;

  // ---- costs + wrenches
  const float4 *sph = c.spheres(h);
  float *wr = c.wrench + (size_t)h * c.wl;
  float cost_pt = 0.0f;
  bool any_grad = false;
  if (a.use_self) {  // reference self_collision_kernel.cuh:19-111 (same loop as the trajectory kernel)
    constexpr int U = 4;
    const int P_pad = (P + 63) & ~63;
    float best = 0.0f;
    int best_k0 = 0x7fffffff;
    for (int k0 = lane; k0 < P_pad; k0 += kFkLanes * U) {
      uint32_t ij[U];
#pragma unroll
      for (int u = 0; u < U; u++) ij[u] = c.pairs[k0 + u * kFkLanes];
      float gmax = staged_pair_penetration(sph, ij[0]);
#pragma unroll
      for (int u = 1; u < U; u++) gmax = fmaxf(gmax, staged_pair_penetration(sph, ij[u]));
      if (gmax > best) { best = gmax; best_k0 = k0; }
    }
    float m = row16_max(best);
    if (m > 0.0f) {
      int best_k = 0x7fffffff;
      if (best == m) {
        float f_best = 0.0f;
#pragma unroll
        for (int u = 0; u < U; u++) {
          const float f = staged_pair_penetration(sph, c.pairs[best_k0 + u * kFkLanes]);
          if (f > f_best) { f_best = f; best_k = best_k0 + u * kFkLanes; }
        }
        best = f_best;
      } else {
        best = 0.0f;
      }
      m = row16_max(best);
      const int kmin = row16_min((best == m && best > 0.0f) ? best_k : 0x7fffffff);
      if (kmin != 0x7fffffff && m > 0.0f) {
        any_grad = true;
        if (lane == 0) cost_pt += self_pair_apply(c, h, m, kmin);
      }
    }
  }
  if (a.use_scene) {
    point_link_masks<0, KINDS>(c, a.sc, h, lane);
    for (int s0 = 0; s0 < S; s0 += kFkLanes) {
      const int s = s0 + lane;
      float d = 0.0f;
      f3 g = make_f3(0.f, 0.f, 0.f);
      float4 c4 = make_float4(0.f, 0.f, 0.f, -1.f);
      const uint32_t mask = s < S ? __float_as_uint(wr[c.sph_link[s] * kWrench + 6]) : 0u;
      if (s < S && (mask != 0u || n_rec > 32)) c4 = scene_sphere<0, KINDS>(c, a.sc, h, s, d, g, mask);
      cost_pt += d;
      any_grad = wrench_add_serialised(c, h, s, make_f3(c4.x, c4.y, c4.z), g, lane64) || any_grad;
    }
  }
  // tool-pose goal-set cost (wp_tool_pose.py:456-692), one tool frame per lane
  point_tool_pose(c, ia.tp, ia.tool_frame_map, T, n, 0, h, (size_t)n * T, ia.out_link_pos, ia.out_link_quat, lane, lane64,
                  cost_pt, any_grad);
  // c-space bound cost (wp_cspace_position.py:232-362): its gradient is already in joint space
  float gp_joint[kDofIters];
#pragma unroll
  for (int it = 0; it < kDofIters; it++) {
    const int d = it * kFkLanes + lane;
    float g = 0.0f;
    if (d < D) {
      float pl = ia.p_b[d], pu = ia.p_b[D + d];
      { const float r = pu - pl, eta_p = ia.cs_eta[0]; pl = pl + eta_p * r; pu = pu - eta_p * r; }
      const float cc = cspace_bound_term(c.q[h * D + d], pl, pu, ia.cs_weight[0], g);
      cost_pt += cc;
      if (ia.out_cspace_cost) ia.out_cspace_cost[(size_t)n * D + d] = cc;
    }
    gp_joint[it] = g;
  }
  cost_pt = row16_sum(cost_pt);
  if (lane == 0) a.out_cost[n] = cost_pt;
  point_vjp_gather(c, h, any_grad, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < kDofIters; it++) {
    const int d = it * kFkLanes + lane;
    if (d < D) ia.out_grad_q[(size_t)n * D + d] = c.q[h * D + d] + gp_joint[it];
  }
}

// ---- compile-time shapes (fused_shapes.hpp): one launcher per shape, each in its own translation unit.
// Returns 1 when the shape holds an instantiation for exactly these arguments and it was launched (*err = what the attribute
// call said), 0 otherwise (the caller runs the generic kernel).
// the launch form a PLAIN shape assumes (fused_shapes.hpp)
static bool fused_plain_launch(const FusedTrajArgs &a) {
  return a.use_self == 1 && a.use_scene == 1 && a.enable_speed_metric == 1 && a.out_position == nullptr && a.out_spheres == nullptr &&
         a.prof == nullptr && a.use_multi_env == 0 && a.num_envs == 1 && a.scene_rows == 0 && a.dispatch_ws != nullptr &&
         a.sphere_padding != nullptr;
}
// ... and of its TERMS instantiation: a trajectory-optimisation iteration (pose + c-space STATE, no torque limits, no cost outputs)
static bool fused_plain_terms(const FusedTrajArgs &a) {
  return a.use_pose == 1 && a.use_cspace == 1 && a.use_torque == 0 && a.tp.out_distance == nullptr && a.tp.out_position_distance == nullptr &&
         a.tp.out_rotation_distance == nullptr && a.tp.out_goalset_idx == nullptr && a.cs.out_cost == nullptr;
}
template <class SH>
static bool fused_shape_matches(const FusedTrajArgs &a, int threads) {
  if (SH::kPlain && !fused_plain_launch(a)) return false;
  return a.bs.padded_horizon == SH::kH && a.bs.n_knots == SH::kNK && a.bs.dof == SH::kD && a.nlinks == SH::kL && a.nspheres == SH::kS &&
         a.npairs == SH::kP && a.chain_len == SH::kC && a.lane_lists != nullptr && a.lane_len0 == SH::kLen0 && a.lane_len1 == SH::kLen1 &&
         threads == SH::kNT && (SH::kNCub < 0 || (a.sc.max_cuboids == SH::kNCub && a.sc.max_voxel_grids == SH::kNVox));
}
#define CUROBO_FUSED_SHAPE_LAUNCHER_PARAMS \
  const FusedTrajArgs &a, int deg, int sweep, int kinds, bool terms, int batch, int threads, size_t lds, hipStream_t st, hipError_t *err
#define CUROBO_FUSED_DECLARE_SHAPE(ID) int fused_shape_launch_##ID(CUROBO_FUSED_SHAPE_LAUNCHER_PARAMS);
CUROBO_FUSED_FOR_EACH_SHAPE(CUROBO_FUSED_DECLARE_SHAPE)
#undef CUROBO_FUSED_DECLARE_SHAPE

#if CUROBO_FUSED_SHAPE_TU > 0
#define CUROBO_FUSED_CAT2(a, b) a##b
#define CUROBO_FUSED_CAT(a, b) CUROBO_FUSED_CAT2(a, b)
#define CUROBO_FUSED_CAT3_(a, b, c) a##b##c
#define CUROBO_FUSED_CAT3(a, b, c) CUROBO_FUSED_CAT3_(a, b, c)
int CUROBO_FUSED_CAT(fused_shape_launch_, CUROBO_FUSED_SHAPE_TU)(CUROBO_FUSED_SHAPE_LAUNCHER_PARAMS) {
  using SH = CUROBO_FUSED_CAT(CUROBO_FUSED_SHAPE_, CUROBO_FUSED_SHAPE_TU);
  if (!fused_shape_matches<SH>(a, threads)) return 0;
#define CUROBO_FUSED_SHAPE_KERNEL(DG, SW, KD, TM)                                                                      \
  if (deg == DG && sweep == SW && kinds == KD && terms == TM && (!(TM) || !SH::kPlain || fused_plain_terms(a))) {      \
    auto kfn = rollout_trajectory_fused_kernel<DG, SW, KD, TM, SH>;                                                    \
    if (batch <= 0) return 1; /* query only */                                                                         \
    *err = lds > 64 * 1024 ? hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess; \
    if (*err == hipSuccess) hipLaunchKernelGGL(kfn, dim3((unsigned)batch), dim3(threads), lds, st, a);                 \
    return 1;                                                                                                          \
  }
  CUROBO_FUSED_CAT3(CUROBO_FUSED_SHAPE_, CUROBO_FUSED_SHAPE_TU, _KERNELS)(CUROBO_FUSED_SHAPE_KERNEL)
#undef CUROBO_FUSED_SHAPE_KERNEL
  return 0;
}
#ifdef CUROBO_FUSED_JIT_SHAPE
// entry points of a run-time compiled shape object (loaded by curobo_amd/backends/fused_jit.py, handed to
// curobo_hip_rollout_fused_register_shape): the launcher behind a C signature, and the size of the argument block so that an
// object built from other sources than the library's is refused
extern "C" __attribute__((visibility("default"))) int curobo_fused_jit_launch(const void *args, int deg, int sweep, int kinds, int terms,
                                                                               int batch, int threads, size_t lds, void *stream, int *err) {
  hipError_t e = hipSuccess;
  const int r = CUROBO_FUSED_CAT(fused_shape_launch_, CUROBO_FUSED_SHAPE_TU)(*static_cast<const FusedTrajArgs *>(args), deg, sweep, kinds,
                                                                             terms != 0, batch, threads, lds, (hipStream_t)stream, &e);
  *err = (int)e;
  return r;
}
extern "C" __attribute__((visibility("default"))) int curobo_fused_jit_args_bytes(void) { return (int)sizeof(FusedTrajArgs); }
#endif
#endif

}  // namespace curobo_hip

#if CUROBO_FUSED_SHAPE_TU == 0  // ---- the C ABI and the generic instantiations: main translation unit only
using namespace curobo_hip;

static long long *g_fused_prof = nullptr;
static bool g_fused_shapes_enabled = true;
// shapes compiled at run time (curobo_fused_jit_launch of their objects), tried before the built-in table
typedef int (*fused_jit_launcher_t)(const void *, int, int, int, int, int, int, size_t, void *, int *);
// A fixed table with an atomic count: ctypes releases the GIL around the entry points, so one thread may register a shape
// while another launches -- a launch reads the count once (acquire) and only the slots below it, a registration writes its slot
// before it publishes the count (release) under a lock that serialises writers (the std::vector here could be
// reallocated under a reader).
constexpr int kMaxJitShapes = 64;
static fused_jit_launcher_t g_jit_shapes[kMaxJitShapes];
static std::atomic<int> g_jit_shape_count{0};
static std::mutex g_jit_shape_lock;
struct JitShapeView {
  int n;
  const fused_jit_launcher_t *begin() const { return g_jit_shapes; }
  const fused_jit_launcher_t *end() const { return g_jit_shapes + n; }
  size_t size() const { return (size_t)n; }
  fused_jit_launcher_t operator[](size_t i) const { return g_jit_shapes[i]; }
};
static JitShapeView fused_jit_shapes() { return JitShapeView{g_jit_shape_count.load(std::memory_order_acquire)}; }
constexpr int kJitShapeIdBase = 100;
CUROBO_EXPORT int curobo_hip_rollout_fused_register_shape(void *launcher, int args_bytes) {
  CUROBO_REQUIRE(launcher != nullptr, "rollout_fused_register_shape: NULL launcher%s", "");
  CUROBO_REQUIRE(args_bytes == (int)sizeof(FusedTrajArgs),
                 "rollout_fused_register_shape: the shape object was built for an argument block of %d bytes, this library's is %d "
                 "(compile it from this library's sources)", args_bytes, (int)sizeof(FusedTrajArgs));
  std::lock_guard<std::mutex> guard(g_jit_shape_lock);
  const int n = g_jit_shape_count.load(std::memory_order_relaxed);
  for (int i = 0; i < n; i++)
    if ((void *)g_jit_shapes[i] == launcher) return CUROBO_HIP_OK;
  CUROBO_REQUIRE(n < kMaxJitShapes, "rollout_fused_register_shape: more than %d run-time shapes", kMaxJitShapes);
  g_jit_shapes[n] = reinterpret_cast<fused_jit_launcher_t>(launcher);
  g_jit_shape_count.store(n + 1, std::memory_order_release);
  return CUROBO_HIP_OK;
}
CUROBO_EXPORT int curobo_hip_rollout_fused_set_shapes_enabled(int enabled) {
  g_fused_shapes_enabled = enabled != 0;
  return CUROBO_HIP_OK;
}

// development hook (not part of the drop-in surface): device buffer [batch][8] of int64 that
// receives 100 MHz wall-clock stamps at the phase boundaries of every fused launch; NULL = off
CUROBO_EXPORT int curobo_hip_rollout_fused_set_profile_buffer(int64_t *device_buffer) {
  g_fused_prof = (long long *)device_buffer;
  return CUROBO_HIP_OK;
}

// Same hook for a SEQUENCE of launches (e.g. the launches recorded into a hipGraph): launch k after
// this call stamps into block k = device_buffer + k * block_rows * 16 (launches beyond n_blocks, or
// with more than block_rows trajectories, are not stamped).  NULL / 0 ends the sequence.
static long long *g_fused_prof_seq = nullptr;
static int g_fused_prof_blocks = 0, g_fused_prof_rows = 0, g_fused_prof_next = 0;
CUROBO_EXPORT int curobo_hip_rollout_fused_set_profile_sequence(int64_t *device_buffer, int n_blocks, int block_rows) {
  g_fused_prof_seq = (long long *)device_buffer;
  g_fused_prof_blocks = device_buffer ? n_blocks : 0;
  g_fused_prof_rows = block_rows;
  g_fused_prof_next = 0;
  return CUROBO_HIP_OK;
}

// workgroup size and LDS layout of the trajectory kernels (the layout depends on the number of waves
// through the per-wave scene lists): 8 waves when two workgroups then fit in a CU's LDS, see the
// kernel's header; else one row per point up to 16 waves
static FusedLayout trajectory_launch_shape(int H, int D, int L, int S, int C, int P, int n_rec, int n_dyn, int *threads_out,
                                           int rings = 1, int n_lane = 0) {
  int threads = ((H * kFkLanes + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  if (threads > 512 && (size_t)fused_layout(H, D, L, S, C, P, n_rec, n_dyn, 8, rings, n_lane).total * sizeof(float) <= 80 * 1024) threads = 512;
  static const int force_threads = [] { const char *e = getenv("CUROBO_HIP_FUSED_THREADS"); return e ? atoi(e) : 0; }();
  if (force_threads >= 64 && force_threads <= 1024) threads = force_threads & ~63;  // tuning knob
  *threads_out = threads;
  return fused_layout(H, D, L, S, C, P, n_rec, n_dyn, threads >> 6, rings, n_lane);
}

CUROBO_EXPORT int curobo_hip_rollout_trajectory_fused_lds_bytes(int padded_horizon, int dof, int num_links,
                                                                   int num_spheres, int num_collision_pairs,
                                                                   int link_chain_len, int num_obstacles) {
  int threads;
  const FusedLayout lay = trajectory_launch_shape(padded_horizon, dof, num_links, num_spheres, link_chain_len,
                                                  num_collision_pairs, num_obstacles, 0, &threads, 2);
  return lay.total * (int)sizeof(float);
}

// layout and workgroup size of a launch; drops the lane form of the pair pass (a.lane_lists) where it would cost a workgroup slot
static FusedLayout fused_resolve_layout(FusedTrajArgs &a, int n_rec, bool with_terms, int *threads_out) {
  const int H = a.bs.padded_horizon, D = a.bs.dof, L = a.nlinks, S = a.nspheres, C = a.chain_len, P = a.npairs;
  const int n_dyn = a.use_cspace ? 4 * H * D : 0, rings = with_terms ? 1 : 2;
  int threads;
  FusedLayout lay = trajectory_launch_shape(H, D, L, S, C, P, n_rec, n_dyn, &threads, rings, a.lane_lists ? (a.lane_len0 + a.lane_len1) * 64 : 0);
  if (a.lane_lists) {
    // the lane form must not cost a workgroup slot: where its lists (64 words per entry, whatever the sphere count) push
    // the trajectory over 160 KB, or over the 80 KB that let two workgroups share a CU, the pass walks pair_locations
    int threads0;
    const FusedLayout lay0 = trajectory_launch_shape(H, D, L, S, C, P, n_rec, n_dyn, &threads0, rings, 0);
    const size_t l1 = (size_t)lay.total * sizeof(float), l0 = (size_t)lay0.total * sizeof(float);
    if (l1 > 160 * 1024 || (l1 > 80 * 1024 && l0 <= 80 * 1024) || (a.use_torque && !fused_torque_fits(lay, H, D, L, S))) {
      a.lane_lists = nullptr; a.lane_len0 = 0; a.lane_len1 = 0;
      lay = lay0;
      threads = threads0;
    }
  }
  *threads_out = threads;
  return lay;
}

// workgroup size a launch with these dimensions uses (a compile-time shape holds it as a constant).  Returns the thread count.
CUROBO_EXPORT int curobo_hip_rollout_fused_threads(int padded_horizon, int dof, int num_links, int num_spheres, int num_collision_pairs,
                                                   int link_chain_len, int self_lane_len, int num_obstacles, int with_trajopt_terms) {
  FusedTrajArgs a{};
  a.bs.padded_horizon = padded_horizon; a.bs.dof = dof; a.nlinks = num_links; a.nspheres = num_spheres; a.npairs = num_collision_pairs;
  a.chain_len = link_chain_len; a.lane_len0 = self_lane_len & 0xffff; a.lane_len1 = (self_lane_len >> 16) & 0xffff;
  static const uint32_t some_list = 0u;
  a.lane_lists = self_lane_len ? &some_list : nullptr;
  a.use_cspace = with_trajopt_terms ? 1 : 0;
  int threads = 0;
  (void)fused_resolve_layout(a, num_obstacles, with_trajopt_terms != 0, &threads);
  return threads;
}

CUROBO_EXPORT int curobo_hip_rollout_fused_shape_id(int padded_horizon, int n_knots, int dof, int num_links, int num_spheres,
                                                    int num_collision_pairs, int link_chain_len, int self_lane_len, int max_cuboids,
                                                    int max_voxel_grids, int bspline_degree, int sweep_steps, int kinds,
                                                    int with_trajopt_terms, int plain_launch) {
#ifdef CUROBO_FUSED_ONLY_C2
  return 0;
#else
  FusedTrajArgs a{};
  a.bs.padded_horizon = padded_horizon; a.bs.n_knots = n_knots; a.bs.dof = dof; a.nlinks = num_links; a.nspheres = num_spheres;
  a.npairs = num_collision_pairs; a.chain_len = link_chain_len; a.sc.max_cuboids = max_cuboids; a.sc.max_voxel_grids = max_voxel_grids;
  a.lane_len0 = self_lane_len & 0xffff; a.lane_len1 = (self_lane_len >> 16) & 0xffff;
  static const uint32_t some_list = 0u;
  static const float some_float = 0.0f;
  static int32_t some_ws = 0;
  a.lane_lists = self_lane_len ? &some_list : nullptr;
  if (plain_launch) {  // self + scene collision with the speed metric, one environment, longest-first dispatch, nothing materialised
    a.use_self = 1; a.use_scene = 1; a.enable_speed_metric = 1; a.num_envs = 1; a.dispatch_ws = &some_ws; a.sphere_padding = &some_float;
  }
  a.use_cspace = with_trajopt_terms ? 1 : 0;
  a.use_pose = (with_trajopt_terms && plain_launch) ? 1 : 0;  // (plain + terms = a trajectory-optimisation iteration: pose + c-space STATE)
  int threads;
  (void)fused_resolve_layout(a, max_cuboids + max_voxel_grids, with_trajopt_terms != 0, &threads);
  {
    const JitShapeView v = fused_jit_shapes();
    for (size_t i = 0; i < v.size(); i++) {
      int e2 = 0;
      if (v[i](&a, bspline_degree, sweep_steps, kinds, with_trajopt_terms != 0 ? 1 : 0, 0, threads, 0, nullptr, &e2)) return kJitShapeIdBase + (int)i;
    }
  }
  hipError_t e = hipSuccess;
#define CUROBO_FUSED_QUERY_SHAPE(ID) \
  if (fused_shape_launch_##ID(a, bspline_degree, sweep_steps, kinds, with_trajopt_terms != 0, 0, threads, 0, nullptr, &e)) return ID;
  CUROBO_FUSED_FOR_EACH_SHAPE(CUROBO_FUSED_QUERY_SHAPE)
#undef CUROBO_FUSED_QUERY_SHAPE
  return 0;
#endif
}

static int rollout_trajectory_fused_impl(
    const char *what, const curobo_hip_trajopt_terms *terms,
    float *out_cost, float *out_grad_knots, float *out_position, float *out_robot_spheres, const float *u_position,
    const float *start_position, const float *start_velocity, const float *start_acceleration, const float *start_jerk,
    const float *goal_position, const float *goal_velocity, const float *goal_acceleration, const float *goal_jerk,
    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt, const uint8_t *use_implicit_goal_state,
    const float *fixed_transform, const float *robot_spheres, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const float *joint_offset_map, const float *sphere_padding,
    const float *self_collision_weight, const int16_t *pair_locations, const curobo_hip_scene *scene,
    const float *scene_collision_weight, const float *activation_distance, const float *speed_dt,
    const int32_t *env_query_idx, int num_envs, int use_multi_env, int batch_size, int padded_horizon, int dof,
    int n_knots, int bspline_degree, int num_links, int num_spheres, int num_collision_pairs, int link_chain_len,
    int sweep_steps, int enable_speed_metric, int32_t *dispatch_ws, int dispatch_phase, const uint32_t *self_lane_lists,
    int self_lane_len, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(bspline_degree >= 3 && bspline_degree <= 5, "%s: bspline_degree must be 3, 4 or 5", what);
  CUROBO_REQUIRE(sweep_steps == 0 || sweep_steps == 3, "%s: sweep_steps must be 0 or 3", what);
  CUROBO_REQUIRE(num_links >= 1 && num_links <= 128 && dof >= 1 && padded_horizon >= 2 && n_knots >= 1,
                 "%s: bad dimensions", what);
  CUROBO_REQUIRE(num_spheres < 4096, "%s: at most 4095 spheres", what);
  CUROBO_REQUIRE(link_chain_len >= 1, "%s: link_chain_len must be >= 1", what);
  CUROBO_REQUIRE(((uintptr_t)pair_locations & 3) == 0, "%s: pair_locations must be 4-byte aligned", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  FusedTrajArgs a{};
  a.out_cost = out_cost; a.out_grad_knots = out_grad_knots; a.out_position = out_position;
  a.out_spheres = out_robot_spheres;
  a.bs.u = u_position;
  a.bs.start[0] = start_position; a.bs.start[1] = start_velocity; a.bs.start[2] = start_acceleration; a.bs.start[3] = start_jerk;
  a.bs.goal[0] = goal_position; a.bs.goal[1] = goal_velocity; a.bs.goal[2] = goal_acceleration; a.bs.goal[3] = goal_jerk;
  a.bs.start_idx = start_idx; a.bs.goal_idx = goal_idx; a.bs.traj_dt = traj_dt; a.bs.use_implicit_goal = use_implicit_goal_state;
  a.bs.batch = batch_size; a.bs.padded_horizon = padded_horizon; a.bs.dof = dof; a.bs.n_knots = n_knots;
  a.fixed_transform = fixed_transform; a.robot_spheres = robot_spheres; a.joint_offset = joint_offset_map;
  a.joint_map_type = joint_map_type; a.joint_map = joint_map; a.link_map = link_map; a.link_sphere_map = link_sphere_map;
  a.link_chain_data = link_chain_data; a.link_chain_offsets = link_chain_offsets;
  a.sphere_padding = sphere_padding; a.w_self = self_collision_weight; a.pairs = pair_locations;
  a.use_self = (pair_locations && self_collision_weight && num_collision_pairs > 0) ? 1 : 0;
  a.use_scene = (scene && scene_collision_weight) ? 1 : 0;
  if (scene) a.sc = *scene;
  a.w_scene = scene_collision_weight; a.eta = activation_distance; a.speed_dt = speed_dt; a.env_query_idx = env_query_idx;
  a.batch = batch_size; a.nlinks = num_links; a.nspheres = num_spheres; a.npairs = a.use_self ? num_collision_pairs : 0;
  a.chain_len = link_chain_len; a.dpad = dof | 1; a.num_envs = num_envs; a.use_multi_env = use_multi_env;
  a.enable_speed_metric = (a.use_scene && enable_speed_metric) ? 1 : 0;
  a.prof = g_fused_prof;
  if (g_fused_prof_seq && g_fused_prof_next < g_fused_prof_blocks && batch_size <= g_fused_prof_rows)
    a.prof = g_fused_prof_seq + (size_t)g_fused_prof_next++ * g_fused_prof_rows * 16;
  a.dispatch_ws = dispatch_ws; a.dispatch_phase = dispatch_phase;
  static const bool no_lanes = getenv("CUROBO_HIP_SELF_ROWS") != nullptr;  // development knob: the row form of the pair pass
  if (self_lane_lists && a.use_self && !no_lanes) {
    a.lane_lists = self_lane_lists; a.lane_len0 = self_lane_len & 0xffff; a.lane_len1 = (self_lane_len >> 16) & 0xffff;
    CUROBO_REQUIRE(a.lane_len0 + a.lane_len1 > 0 && num_spheres <= 128 && num_collision_pairs < 65535,
                   "%s: self_lane_lists does not describe this robot (curobo_hip_self_lane_lists_host)", what);
  }
  static const int scene_rows = [] { const char *e = getenv("CUROBO_HIP_SCENE_ROWS"); return e ? atoi(e) : 0; }();
  a.scene_rows = scene_rows;
  CUROBO_REQUIRE(!dispatch_ws || dispatch_phase == 0 || dispatch_phase == 1, "%s: dispatch_phase must be 0 or 1", what);
  CUROBO_REQUIRE(!a.enable_speed_metric || speed_dt, "%s: speed metric needs speed_dt", what);
  const int n_rec = a.use_scene ? a.sc.max_cuboids + a.sc.max_voxel_grids : 0;
  if (!a.use_scene) { a.sc.max_cuboids = 0; a.sc.max_voxel_grids = 0; }
  if (terms) {
    const curobo_hip_trajopt_terms &t = *terms;
    a.use_pose = (t.n_tool_frames > 0 && t.goal_position) ? 1 : 0;
    a.use_cspace = t.cspace_weight ? 1 : 0;
    // development knob (timing by elimination): CUROBO_HIP_TERMS_MASK bit 0 = tool pose, bit 1 = c-space STATE
    static const int terms_mask = [] { const char *e = getenv("CUROBO_HIP_TERMS_MASK"); return e ? atoi(e) : 3; }();
    if (!(terms_mask & 1)) a.use_pose = 0;
    if (!(terms_mask & 2)) a.use_cspace = 0;
    if (a.use_pose) {
      CUROBO_REQUIRE(t.tool_frame_map && t.goal_quat && t.idxs_goal && t.position_orientation_weight &&
                         t.terminal_pose_axes_weight_factor && t.non_terminal_pose_axes_weight_factor &&
                         t.terminal_pose_convergence_tolerance && t.non_terminal_pose_convergence_tolerance &&
                         t.project_distance_to_goal && t.num_goalset >= 1,
                     "%s: incomplete tool-pose terms", what);
      CUROBO_REQUIRE(t.rotation_method >= 0 && t.rotation_method <= 2, "%s: rotation_method must be 0, 1 or 2", what);
      ToolPoseArgs &tp = a.tp;
      tp.out_distance = t.out_pose_distance; tp.out_position_distance = t.out_position_distance;
      tp.out_rotation_distance = t.out_rotation_distance; tp.out_goalset_idx = t.out_goalset_idx;
      tp.goal_position = t.goal_position; tp.goal_quat = t.goal_quat; tp.idxs_goal = t.idxs_goal;
      tp.position_orientation_weight = t.position_orientation_weight;
      tp.terminal_axes_weight = t.terminal_pose_axes_weight_factor;
      tp.non_terminal_axes_weight = t.non_terminal_pose_axes_weight_factor;
      tp.terminal_tolerance = t.terminal_pose_convergence_tolerance;
      tp.non_terminal_tolerance = t.non_terminal_pose_convergence_tolerance;
      tp.project_distance_to_goal = t.project_distance_to_goal;
      tp.batch = batch_size; tp.horizon = padded_horizon; tp.num_links = t.n_tool_frames; tp.num_goalset = t.num_goalset;
      tp.rotation_method = t.rotation_method;
      a.tool_frame_map = t.tool_frame_map; a.n_tool_frames = t.n_tool_frames;
    }
    if (a.use_cspace) {
      CUROBO_REQUIRE(t.state_dt && t.p_b && t.v_b && t.a_b && t.j_b && t.effort_b && t.cspace_activation_distance &&
                         t.squared_l2_regularization_weights && t.cspace_target_weight &&
                         t.cspace_non_terminal_weight_factor && t.cspace_target_dof_weight && t.target_joint_position &&
                         t.idxs_target_joint_position,
                     "%s: incomplete c-space terms", what);
      CUROBO_REQUIRE(dof <= 64, "%s: c-space term supports dof <= 64", what);
      CspaceStateArgs &cs = a.cs;
      cs.out_cost = t.out_cspace_cost; cs.state_dt = t.state_dt; cs.target = t.target_joint_position;
      cs.idxs_target = t.idxs_target_joint_position; cs.p_b = t.p_b; cs.v_b = t.v_b; cs.a_b = t.a_b; cs.j_b = t.j_b;
      cs.effort_b = t.effort_b; cs.weight = t.cspace_weight; cs.activation_distance = t.cspace_activation_distance;
      cs.sql2_weights = t.squared_l2_regularization_weights; cs.target_weight = t.cspace_target_weight;
      cs.non_terminal_factor = t.cspace_non_terminal_weight_factor; cs.target_dof_weight = t.cspace_target_dof_weight;
      cs.write_grad = 1; cs.batch = batch_size; cs.horizon = padded_horizon; cs.dof = dof;
      cs.retime_weights = t.retime_weights; cs.retime_reg_weights = t.retime_regularization_weights;
      if (t.use_torque_limits) {
        CUROBO_REQUIRE(t.link_masses_com && t.link_inertias && t.gravity && t.level_links,
                       "%s: torque limits need link_masses_com, link_inertias, gravity, level_links", what);
        a.link_masses_com = t.link_masses_com; a.link_inertias = t.link_inertias; a.gravity = t.gravity;
        a.level_links = t.level_links; a.use_torque = 1;
      }
    }
  }
  int threads;
  static const bool force_terms = getenv("CUROBO_HIP_FORCE_TERMS") != nullptr;
  const bool with_terms = a.use_pose || a.use_cspace || force_terms;  // the TERMS instantiation: one scene ring per wave
  const FusedLayout lay = fused_resolve_layout(a, n_rec, with_terms, &threads);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CUROBO_REQUIRE(lds <= 160 * 1024, "%s: trajectory does not fit in LDS (%zu bytes); use the unfused kernels", what, lds);
  CUROBO_REQUIRE(!a.use_torque || fused_torque_fits(lay, padded_horizon, dof, num_links, num_spheres),
                 "%s: the inverse-dynamics state of this robot does not fit the LDS regions it borrows; run the torque "
                 "limits on the kernel sequence (curobo_hip_rollout_trajopt_fused_torque_fits)", what);
  // scenes with analytic primitives in the cuboid store run the one instantiation that tests the tag (KINDS = 7)
  const int kinds = (a.sc.max_cuboids > 0 && a.sc.cuboid_has_primitives) ? 7 : ((a.sc.max_cuboids > 0 ? 1 : 0) | (a.sc.max_voxel_grids > 0 ? 2 : 0));
  hipStream_t st = (hipStream_t)stream;
#ifdef CUROBO_FUSED_ONLY_C2  // (experiment builds: the C2 instantiation with its compile-time shape, nothing else)
#define CUROBO_FUSED_KERNEL_OF(DG, SW, KD) rollout_trajectory_fused_kernel<DG, SW, KD, false, CUROBO_FUSED_SHAPE_1>
#else
#define CUROBO_FUSED_KERNEL_OF(DG, SW, KD) \
  (with_terms ? rollout_trajectory_fused_kernel<DG, SW, KD, true> : rollout_trajectory_fused_kernel<DG, SW, KD, false>)
#endif
#define CUROBO_FUSED_LAUNCH(DG, SW, KD)                                                                        \
  do {                                                                                                         \
    auto kfn = CUROBO_FUSED_KERNEL_OF(DG, SW, KD);                                                             \
    if (lds > 64 * 1024) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: cannot raise LDS limit: %s", what, hipGetErrorString(e)); \
    }                                                                                                          \
    static const bool dbg = getenv("CUROBO_HIP_FUSED_DEBUG") != nullptr;                                      \
    if (dbg) {                                                                                                 \
      int nb = -1;                                                                                             \
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kfn, threads, lds);                      \
      fprintf(stderr, "[curobo_hip] fused: threads %d lds %zu -> %d workgroups/CU\n", threads, lds, nb); \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, dim3((unsigned)batch_size), dim3(threads), lds, st, a);                            \
  } while (0)
#ifndef CUROBO_FUSED_ONLY_C2
  {  // a compile-time shape that matches every dimension of this launch (fused_shapes.hpp), most specific first
    static const bool env_off = getenv("CUROBO_HIP_FUSED_NO_SHAPES") != nullptr;  // development knob: always the generic kernel
    const bool no_shapes = env_off || !g_fused_shapes_enabled;
    hipError_t se = hipSuccess;
    if (!no_shapes) {  // shapes compiled at run time for this robot / horizon
      for (fused_jit_launcher_t fn : fused_jit_shapes()) {
        int e = 0;
        if (fn(&a, bspline_degree, sweep_steps, kinds, with_terms ? 1 : 0, batch_size, threads, lds, (void *)st, &e)) {
          if (e != 0) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: cannot raise LDS limit: %s", what, hipGetErrorString((hipError_t)e));
          return check_launch(what, st);
        }
      }
    }
#define CUROBO_FUSED_TRY_SHAPE(ID)                                                                                              \
    if (!no_shapes && fused_shape_launch_##ID(a, bspline_degree, sweep_steps, kinds, with_terms, batch_size, threads, lds, st, &se)) { \
      if (se != hipSuccess) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: cannot raise LDS limit: %s", what, hipGetErrorString(se));  \
      return check_launch(what, st);                                                                                            \
    }
    CUROBO_FUSED_FOR_EACH_SHAPE(CUROBO_FUSED_TRY_SHAPE)
#undef CUROBO_FUSED_TRY_SHAPE
  }
#endif
#ifdef CUROBO_FUSED_ONLY_C2  // experiment builds (tools/r05/build_fused_variant.sh): one instantiation, seconds to compile
  CUROBO_REQUIRE(bspline_degree == 3 && sweep_steps == 3 && kinds == 1 && !with_terms, "%s: this experiment build only holds the C2 instantiation", what);
  CUROBO_FUSED_LAUNCH(3, 3, 1);
  return check_launch(what, st);
#else
#define CUROBO_FUSED_KINDS(DG, SW)                         \
  do {                                                     \
    if (kinds == 2) CUROBO_FUSED_LAUNCH(DG, SW, 2);        \
    else if (kinds == 3) CUROBO_FUSED_LAUNCH(DG, SW, 3);   \
    else if (kinds == 7) CUROBO_FUSED_LAUNCH(DG, SW, 7);   \
    else CUROBO_FUSED_LAUNCH(DG, SW, 1);                   \
  } while (0)
#define CUROBO_FUSED_SWEEP(DG)                             \
  do {                                                     \
    if (sweep_steps == 0) CUROBO_FUSED_KINDS(DG, 0);       \
    else CUROBO_FUSED_KINDS(DG, 3);                        \
  } while (0)
  if (bspline_degree == 3) CUROBO_FUSED_SWEEP(3);
  else if (bspline_degree == 4) CUROBO_FUSED_SWEEP(4);
  else CUROBO_FUSED_SWEEP(5);
#undef CUROBO_FUSED_SWEEP
#undef CUROBO_FUSED_KINDS
#undef CUROBO_FUSED_LAUNCH
  return check_launch(what, st);
#endif
}

__global__ void dispatch_ws_init_kernel(int32_t *ws, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { ws[i] = i; ws[B + i] = i; ws[2 * B + i] = 0; ws[3 * B + i] = 0; }
}

CUROBO_EXPORT int curobo_hip_rollout_dispatch_ws_size(int batch_size) { return batch_size > 0 ? 4 * batch_size : 0; }

CUROBO_EXPORT int curobo_hip_rollout_dispatch_ws_init(int32_t *dispatch_ws, int batch_size, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(dispatch_ws && batch_size > 0, "rollout_dispatch_ws_init: NULL workspace or empty batch%s", "");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dispatch_ws_init_kernel, dim3((batch_size + 255) / 256), dim3(256), 0, st, dispatch_ws, batch_size);
  return check_launch("rollout_dispatch_ws_init", st);
}

#define CUROBO_TRAJ_PARAMS                                                                                          \
  float *out_cost, float *out_grad_knots, float *out_position, float *out_robot_spheres, const float *u_position,     \
      const float *start_position, const float *start_velocity, const float *start_acceleration,                      \
      const float *start_jerk, const float *goal_position, const float *goal_velocity, const float *goal_acceleration, \
      const float *goal_jerk, const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,                \
      const uint8_t *use_implicit_goal_state, const float *fixed_transform, const float *robot_spheres,               \
      const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map, const int16_t *link_sphere_map, \
      const int16_t *link_chain_data, const int16_t *link_chain_offsets, const float *joint_offset_map,               \
      const float *sphere_padding, const float *self_collision_weight, const int16_t *pair_locations,                 \
      const curobo_hip_scene *scene, const float *scene_collision_weight, const float *activation_distance,           \
      const float *speed_dt, const int32_t *env_query_idx, int num_envs, int use_multi_env, int batch_size,           \
      int padded_horizon, int dof, int n_knots, int bspline_degree, int num_links, int num_spheres,                   \
      int num_collision_pairs, int link_chain_len, int sweep_steps, int enable_speed_metric, int32_t *dispatch_ws,    \
      int dispatch_phase, const uint32_t *self_lane_lists, int self_lane_len
#define CUROBO_TRAJ_ARGS                                                                                              \
  out_cost, out_grad_knots, out_position, out_robot_spheres, u_position, start_position, start_velocity,             \
      start_acceleration, start_jerk, goal_position, goal_velocity, goal_acceleration, goal_jerk, start_idx, goal_idx, \
      traj_dt, use_implicit_goal_state, fixed_transform, robot_spheres, joint_map_type, joint_map, link_map,          \
      link_sphere_map, link_chain_data, link_chain_offsets, joint_offset_map, sphere_padding, self_collision_weight,  \
      pair_locations, scene, scene_collision_weight, activation_distance, speed_dt, env_query_idx, num_envs,          \
      use_multi_env, batch_size, padded_horizon, dof, n_knots, bspline_degree, num_links, num_spheres,                \
      num_collision_pairs, link_chain_len, sweep_steps, enable_speed_metric, dispatch_ws, dispatch_phase, self_lane_lists,   \
      self_lane_len

// The pair list dealt to lanes (lane = sphere form of the self-collision pass of the fused kernels): every pair (i, j)
// is given to ONE of its two spheres so that the longest list is as short as possible (a b-matching, found with
// augmenting paths); sphere s is owned by lane s % 64 of pass s / 64.  out_lists_host [(len0 + len1) * 64] words,
// word (e, lane) = partner byte offset (16 j) | pair index << 16, lists in ascending pair index, padded with the NaN
// sphere behind the last one (16 S) | 0xffff << 16.  Returns len0 | len1 << 16 (pass the value on as self_lane_len),
// 0 when this robot is outside the form (more than 128 spheres, 65535 pairs or `capacity_words`), < 0 on bad arguments.
// HOST pointers: call once per robot, copy out_lists_host to the device.
CUROBO_EXPORT int curobo_hip_self_lane_lists_host(uint32_t *out_lists_host, int capacity_words, const int16_t *pair_locations_host,
                                                  int num_collision_pairs, int num_spheres) {
  const char *what = "self_lane_lists_host";
  if (!out_lists_host || !pair_locations_host || num_collision_pairs <= 0 || num_spheres <= 0) {
    set_error(CUROBO_HIP_ERR_INVALID, "%s: bad arguments", what);
    return -1;
  }
  const int P = num_collision_pairs, S = num_spheres;
  if (S > 128 || P >= 65535 || P > 16384) return 0;
  for (int k = 0; k < P; k++) {
    const int i = pair_locations_host[2 * k], j = pair_locations_host[2 * k + 1];
    if (i < 0 || j < 0 || i >= S || j >= S) { set_error(CUROBO_HIP_ERR_INVALID, "%s: pair %d out of range", what, k); return -1; }
  }
  std::vector<int> owner(P), load(S), cap(S), seen(S);
  std::vector<std::vector<int>> owned(S);
  auto other = [&](int k, int x) { const int i = pair_locations_host[2 * k], j = pair_locations_host[2 * k + 1]; return x == i ? j : i; };
  // free one slot of sphere x by moving one of its pairs to that pair's other sphere (recursively)
  std::function<bool(int)> make_room = [&](int x) -> bool {
    for (size_t n = 0; n < owned[x].size(); n++) {
      const int k2 = owned[x][n], y = other(k2, x);
      if (y == x || seen[y]) continue;
      seen[y] = 1;
      if (load[y] < cap[y] || make_room(y)) {
        owned[x].erase(owned[x].begin() + (long)n);
        load[x]--;
        owned[y].push_back(k2);
        load[y]++;
        owner[k2] = y;
        return true;
      }
    }
    return false;
  };
  auto feasible = [&](int c0, int c1) -> bool {
    for (int x = 0; x < S; x++) { cap[x] = x < 64 ? c0 : c1; load[x] = 0; owned[x].clear(); }
    for (int k = 0; k < P; k++) {
      const int e[2] = {pair_locations_host[2 * k], pair_locations_host[2 * k + 1]};
      int x = -1;
      // the emptier of the two spheres when it has room, else make room at either
      const int a0 = load[e[0]] <= load[e[1]] ? e[0] : e[1], a1 = a0 == e[0] ? e[1] : e[0];
      if (load[a0] < cap[a0]) x = a0;
      else if (load[a1] < cap[a1]) x = a1;
      else {
        for (int t = 0; t < 2 && x < 0; t++) {
          std::fill(seen.begin(), seen.end(), 0);
          seen[e[0]] = seen[e[1]] = 1;
          if (cap[e[t]] > 0 && make_room(e[t])) x = e[t];
        }
      }
      if (x < 0) return false;
      owned[x].push_back(k);
      load[x]++;
      owner[k] = x;
    }
    return true;
  };
  int len0 = -1, len1 = -1;
  const int lower = (P + 63) / 64;
  for (int total = lower; total <= P && len0 < 0; total++)
    for (int c1 = 0; c1 <= (S > 64 ? total : 0) && len0 < 0; c1++)
      if (feasible(total - c1, c1)) { len0 = total - c1; len1 = c1; }
  if (len0 < 0 || len0 > 0xffff || len1 > 0xffff || (len0 + len1) * 64 > capacity_words) return 0;
  const uint32_t pad = (uint32_t)(S * 16) | (0xffffu << 16);
  for (int w = 0; w < (len0 + len1) * 64; w++) out_lists_host[w] = pad;
  for (int x = 0; x < S; x++) {
    std::sort(owned[x].begin(), owned[x].end());
    const int base = x < 64 ? 0 : len0, lane = x & 63;
    for (size_t n = 0; n < owned[x].size(); n++)
      out_lists_host[(size_t)(base + (int)n) * 64 + lane] = (uint32_t)(other(owned[x][n], x) * 16) | ((uint32_t)owned[x][n] << 16);
  }
  return len0 | (len1 << 16);
}

CUROBO_EXPORT int curobo_hip_rollout_trajectory_fused(CUROBO_TRAJ_PARAMS, curobo_hip_stream_t stream) {
  return rollout_trajectory_fused_impl("rollout_trajectory_fused", nullptr, CUROBO_TRAJ_ARGS, stream);
}

CUROBO_EXPORT int curobo_hip_rollout_trajopt_fused(CUROBO_TRAJ_PARAMS, const curobo_hip_trajopt_terms *terms,
                                                   curobo_hip_stream_t stream) {
  return rollout_trajectory_fused_impl("rollout_trajopt_fused", terms, CUROBO_TRAJ_ARGS, stream);
}
#undef CUROBO_TRAJ_PARAMS
#undef CUROBO_TRAJ_ARGS

CUROBO_EXPORT int curobo_hip_rollout_trajopt_fused_lds_bytes(int padded_horizon, int dof, int num_links, int num_spheres,
                                                             int num_collision_pairs, int link_chain_len,
                                                             int num_obstacles, int with_cspace_terms) {
  int threads;
  const FusedLayout lay = trajectory_launch_shape(padded_horizon, dof, num_links, num_spheres, link_chain_len, num_collision_pairs,
                                                  num_obstacles, with_cspace_terms ? 4 * padded_horizon * dof : 0, &threads);
  return lay.total * (int)sizeof(float);
}

CUROBO_EXPORT int curobo_hip_rollout_trajopt_fused_torque_fits(int padded_horizon, int dof, int num_links, int num_spheres,
                                                               int num_collision_pairs, int link_chain_len, int num_obstacles) {
  int threads;
  const FusedLayout lay = trajectory_launch_shape(padded_horizon, dof, num_links, num_spheres, link_chain_len, num_collision_pairs,
                                                  num_obstacles, 4 * padded_horizon * dof, &threads, 1);
  return (size_t)lay.total * sizeof(float) <= 160 * 1024 && fused_torque_fits(lay, padded_horizon, dof, num_links, num_spheres) ? 1 : 0;
}

CUROBO_EXPORT int curobo_hip_rollout_ik_fused_lds_bytes(int dof, int num_links, int num_spheres, int num_collision_pairs,
                                                        int link_chain_len, int num_obstacles) {
  const FusedLayout lay = fused_layout(kIkPoints, dof, num_links, num_spheres, link_chain_len, num_collision_pairs, num_obstacles, 0, 0, 1, 0, false);
  return lay.total * (int)sizeof(float);
}

CUROBO_EXPORT int curobo_hip_rollout_ik_fused(
    float *out_cost, float *out_grad_q, float *out_pose_distance, float *out_position_distance,
    float *out_rotation_distance, int32_t *out_goalset_idx, float *out_link_pos, float *out_link_quat,
    float *out_robot_spheres, float *out_cspace_cost, const float *q, const float *goal_position, const float *goal_quat,
    const int32_t *idxs_goal, const float *position_orientation_weight, const float *terminal_pose_axes_weight_factor,
    const float *terminal_pose_convergence_tolerance, const uint8_t *project_distance_to_goal, int num_goalset,
    int rotation_method, const float *p_b, const float *cspace_weight, const float *cspace_activation_distance,
    const float *fixed_transform, const float *robot_spheres, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_sphere_map,
    const int16_t *link_chain_data, const int16_t *link_chain_offsets, const float *joint_offset_map,
    const float *sphere_padding, const float *self_collision_weight, const int16_t *pair_locations,
    const curobo_hip_scene *scene, const float *scene_collision_weight, const float *activation_distance,
    int batch_size, int dof, int num_links, int n_tool_frames, int num_spheres, int num_collision_pairs,
    int link_chain_len, const int32_t *env_query_idx, int num_envs, int use_multi_env, curobo_hip_stream_t stream) {
  const char *what = "rollout_ik_fused";
  CUROBO_REQUIRE((!use_multi_env && num_envs <= 1) || env_query_idx, "%s: per-environment scenes / sphere sets need env_query_idx", what);
  CUROBO_REQUIRE(num_links >= 1 && num_links <= 128 && dof >= 1 && dof <= 64, "%s: bad dimensions", what);
  CUROBO_REQUIRE(n_tool_frames >= 1 && num_goalset >= 1, "%s: need at least one tool frame / goal", what);
  CUROBO_REQUIRE(link_chain_len >= 1, "%s: link_chain_len must be >= 1", what);
  CUROBO_REQUIRE(num_spheres < 4096, "%s: at most 4095 spheres", what);
  CUROBO_REQUIRE(((uintptr_t)pair_locations & 3) == 0, "%s: pair_locations must be 4-byte aligned", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  FusedIkArgs ia{};
  FusedTrajArgs &a = ia.r;
  a.out_cost = out_cost; a.out_spheres = out_robot_spheres;
  a.bs.dof = dof;
  a.fixed_transform = fixed_transform; a.robot_spheres = robot_spheres; a.joint_offset = joint_offset_map;
  a.joint_map_type = joint_map_type; a.joint_map = joint_map; a.link_map = link_map; a.link_sphere_map = link_sphere_map;
  a.link_chain_data = link_chain_data; a.link_chain_offsets = link_chain_offsets;
  a.sphere_padding = sphere_padding; a.w_self = self_collision_weight; a.pairs = pair_locations;
  a.use_self = (pair_locations && self_collision_weight && num_collision_pairs > 0) ? 1 : 0;
  a.use_scene = (scene && scene_collision_weight) ? 1 : 0;
  if (scene) a.sc = *scene;
  if (!a.use_scene) { a.sc.max_cuboids = 0; a.sc.max_voxel_grids = 0; }
  a.w_scene = scene_collision_weight; a.eta = activation_distance;
  a.batch = batch_size; a.nlinks = num_links; a.nspheres = num_spheres; a.npairs = a.use_self ? num_collision_pairs : 0;
  a.chain_len = link_chain_len; a.num_envs = num_envs > 1 ? num_envs : 1; a.use_multi_env = use_multi_env ? 1 : 0;
  a.env_query_idx = env_query_idx;
  ia.x = q; ia.out_grad_q = out_grad_q; ia.tool_frame_map = tool_frame_map; ia.n_tool_frames = n_tool_frames;
  ia.n_points = batch_size; ia.out_link_pos = out_link_pos; ia.out_link_quat = out_link_quat;
  ToolPoseArgs &tp = ia.tp;
  tp.out_distance = out_pose_distance; tp.out_position_distance = out_position_distance;
  tp.out_rotation_distance = out_rotation_distance; tp.out_goalset_idx = out_goalset_idx;
  tp.goal_position = goal_position; tp.goal_quat = goal_quat; tp.idxs_goal = idxs_goal;
  tp.position_orientation_weight = position_orientation_weight;
  tp.terminal_axes_weight = terminal_pose_axes_weight_factor; tp.non_terminal_axes_weight = terminal_pose_axes_weight_factor;
  tp.terminal_tolerance = terminal_pose_convergence_tolerance; tp.non_terminal_tolerance = terminal_pose_convergence_tolerance;
  tp.project_distance_to_goal = project_distance_to_goal;
  tp.batch = batch_size; tp.horizon = 1; tp.num_links = n_tool_frames; tp.num_goalset = num_goalset; tp.rotation_method = rotation_method;
  CUROBO_REQUIRE(rotation_method >= 0 && rotation_method <= 2, "%s: rotation_method must be 0, 1 or 2", what);
  // c-space bound term only (no effort, target or velocity-limited bounds in the IK cost set)
  ia.p_b = p_b; ia.cs_weight = cspace_weight; ia.cs_eta = cspace_activation_distance; ia.out_cspace_cost = out_cspace_cost;
  hipStream_t st = (hipStream_t)stream;
  const int n_rec = a.sc.max_cuboids + a.sc.max_voxel_grids;
  const FusedLayout lay = fused_layout(kIkPoints, dof, num_links, num_spheres, link_chain_len, a.npairs, n_rec, 0, 0, 1, 0, false);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CUROBO_REQUIRE(lds <= 160 * 1024, "%s: 16 configurations do not fit in LDS (%zu bytes); use the unfused kernels", what, lds);
  const int kinds = (a.sc.max_cuboids > 0 && a.sc.cuboid_has_primitives) ? 7 : ((a.sc.max_cuboids > 0 ? 1 : 0) | (a.sc.max_voxel_grids > 0 ? 2 : 0));
  const dim3 grid((unsigned)ceil_div(batch_size, kIkPoints)), block(kIkPoints * kFkLanes);
#define CUROBO_IK_LAUNCH(KD)                                                                                    \
  do {                                                                                                          \
    auto kfn = rollout_ik_fused_kernel<KD>;                                                                     \
    if (lds > 64 * 1024) {                                                                                      \
      hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: cannot raise LDS limit: %s", what, hipGetErrorString(e)); \
    }                                                                                                           \
    hipLaunchKernelGGL(kfn, grid, block, lds, st, ia);                                                          \
  } while (0)
#ifdef CUROBO_FUSED_ONLY_C2
  CUROBO_IK_LAUNCH(1);
#else
  if (kinds == 2) CUROBO_IK_LAUNCH(2);
  else if (kinds == 3) CUROBO_IK_LAUNCH(3);
  else if (kinds == 7) CUROBO_IK_LAUNCH(7);
  else CUROBO_IK_LAUNCH(1);
#endif
#undef CUROBO_IK_LAUNCH
  return check_launch(what, st);
}
#endif  // CUROBO_FUSED_SHAPE_TU == 0
