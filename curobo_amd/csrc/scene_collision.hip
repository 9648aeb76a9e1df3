// scene_collision.hip -- robot spheres vs. world obstacles (oriented cuboids, fp16 ESDF voxel
// grids): activation-shaped penetration cost + world-frame gradient per sphere, optional swept
// (h-1 / h+1) sampling and the CHOMP speed metric.
//
// Reference (NVIDIA Warp, no backend hook): geom/collision/wp_collision_kernel.py:70-166,
// wp_sweep_collision_kernel.py:83-260, wp_speed_metric.py:10-93, wp_collision_common.py:11-96,
// geom/data/data_cuboid.py:461-628, geom/data/data_voxel.py:709-1215, geom/data/helper_pose.py.
//
// gfx950 design: the reference launches one thread per (sphere, obstacle) and float-atomics the
// results into pre-zeroed buffers, once per obstacle type, then a speed-metric kernel.  Here one
// lane owns one sphere of one trajectory point and walks the (few) obstacles itself:
//   * sums are deterministic (obstacle-index order) -- no atomics, no zero_() pass, one launch
//     for every obstacle type, speed metric fused (it only touches the lane's own accumulators);
//   * obstacle records are wave-uniform loads (scalar/L1 broadcast), sphere loads and
//     distance/gradient stores are 16 B per lane, fully coalesced along the sphere axis;
//   * the 128^3 fp16 ESDF (4 MiB) stays L2/Infinity-Cache resident; the 8 corner gathers per
//     query are the cost, not HBM.
#include <cstdlib>

#include "scene_device.hpp"

namespace curobo_hip {

struct SceneArgs {
  float *distance;
  float *gradient;
  const float *spheres;
  curobo_hip_scene sc;
  const float *weight;
  const float *activation_distance;
  const int32_t *env_query_idx;
  const float *speed_dt;
  int batch, horizon, nspheres, use_multi_env, sweep_steps, enable_speed_metric;
};

// KINDS: bit 0 = cuboids present, bit 1 = voxel grids present (separate instantiations keep the
// cuboid-only kernel's register footprint free of the 8-corner voxel gather state)
template <int SWEEP, bool STAGED, int KINDS>
__global__ void __launch_bounds__(256) scene_collision_kernel(const SceneArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const long total = (long)a.batch * a.horizon * a.nspheres;
  const long sidx0 = (long)blockIdx.x * blockDim.x;
  const long sidx = sidx0 + threadIdx.x;
  const int hs = a.horizon * a.nspheres;
  const int n_rec = a.sc.max_cuboids + a.sc.max_voxel_grids;
  const int b_first = (int)(sidx0 / hs);
  ObsRec *recs = reinterpret_cast<ObsRec *>(smem);  // [batch rows touched by this workgroup][n_rec]
  if (STAGED) {
    const long last = (sidx0 + blockDim.x - 1 < total - 1) ? sidx0 + blockDim.x - 1 : total - 1;
    const int nslots = (int)(last / hs) - b_first + 1;
    for (int idx = threadIdx.x; idx < nslots * n_rec; idx += blockDim.x) {
      const int slot = idx / n_rec, o = idx - slot * n_rec;
      const int env = a.use_multi_env ? a.env_query_idx[b_first + slot] : 0;
      recs[idx] = (o < a.sc.max_cuboids) ? load_rec_global<false>(a.sc, env, o)
                                         : load_rec_global<true>(a.sc, env, o - a.sc.max_cuboids);
    }
    __syncthreads();
  }
  if (sidx >= total) return;
  const int b = (int)(sidx / hs);
  const int h = (int)((sidx - (long)b * hs) / a.nspheres);
  const int env = a.use_multi_env ? a.env_query_idx[b] : 0;
  const float *sph_ptr = a.spheres + sidx * 4;
  const float4 s = *reinterpret_cast<const float4 *>(sph_ptr);
  const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;  // neighbours feed the sweep and the speed metric
  const bool has_prev = need_nb && h > 0, has_next = need_nb && h < a.horizon - 1;
  float4 ps = s, ns = s;
  if (has_prev) ps = *reinterpret_cast<const float4 *>(sph_ptr - (size_t)a.nspheres * 4);
  if (has_next) ns = *reinterpret_cast<const float4 *>(sph_ptr + (size_t)a.nspheres * 4);
  float dsum;
  f3 gsum;
  sphere_scene_cost<SWEEP, STAGED, KINDS>(a.sc, recs + (size_t)(b - b_first) * n_rec, env, s, has_prev, ps, has_next, ns,
                                          a.activation_distance[0], a.weight[0], a.enable_speed_metric != 0,
                                          a.enable_speed_metric ? a.speed_dt[0] : 0.0f, dsum, gsum);
  a.distance[sidx] = dsum;
  reinterpret_cast<float4 *>(a.gradient)[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
}


// Packed variant (scenes with ESDF grids, <= 32 obstacle records; cuboid-only scenes keep the in-lane loop, whose
// early reject already leaves little divergent work: measured 38.5 us in-lane vs 45.7 us packed on C2, 109 vs 100 us
// on the C3 ESDF world).  With one sphere per lane and the obstacle loop in-lane, a wavefront
// runs the expensive part -- signed distance + up to six sweep samples, eight fp16 gathers each for an ESDF -- as
// long as ANY of its 64 spheres needs it, and typically a tenth of them do.  Here the workgroup first runs the
// cheap part for every sphere (transform into the obstacle frame + the result-preserving early reject, incl. the
// coarse ESDF minimum), compacts the surviving (sphere, obstacle) items into an LDS queue (sphere-major, obstacle
// ascending), evaluates the queue densely, 256 items per round, and every sphere then adds ITS items in queue
// order -- the same obstacle-index order as the in-lane loop, so the sums are bit-identical to it.
template <int SWEEP, int KINDS>
__global__ void __launch_bounds__(256) scene_collision_packed_kernel(const SceneArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 256;
  const long total = (long)a.batch * a.horizon * a.nspheres;
  const long sidx0 = (long)blockIdx.x * NT;
  const int tid = threadIdx.x, lane64 = tid & 63, wave = tid >> 6;
  const long sidx = sidx0 + tid;
  const int hs = a.horizon * a.nspheres;
  const int n_rec = a.sc.max_cuboids + a.sc.max_voxel_grids;
  const int b_first = (int)(sidx0 / hs);
  const long last = (sidx0 + NT - 1 < total - 1) ? sidx0 + NT - 1 : total - 1;
  const int nslots = (int)(last / hs) - b_first + 1;
  // LDS: records | sphere stash [3][256] float4 | results [256] float4 | prefix [256 + 8] int | queue [256 * n_rec] u16
  ObsRec *recs = reinterpret_cast<ObsRec *>(smem);
  float4 *stash = reinterpret_cast<float4 *>(recs + (size_t)nslots * n_rec);
  float4 *res = stash + 3 * NT;
  int *prefix = reinterpret_cast<int *>(res + NT);
  uint16_t *queue = reinterpret_cast<uint16_t *>(prefix + NT + 8);
  for (int idx = tid; idx < nslots * n_rec; idx += NT) {
    const int slot = idx / n_rec, o = idx - slot * n_rec;
    const int env = a.use_multi_env ? a.env_query_idx[b_first + slot] : 0;
    recs[idx] = (o < a.sc.max_cuboids) ? load_rec_global<false>(a.sc, env, o) : load_rec_global<true>(a.sc, env, o - a.sc.max_cuboids);
  }
  const bool in = sidx < total;
  const int b = in ? (int)(sidx / hs) : b_first;
  const int h = in ? (int)((sidx - (long)b * hs) / a.nspheres) : 0;
  const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;
  const bool has_prev = in && need_nb && h > 0, has_next = in && need_nb && h < a.horizon - 1;
  float4 s = make_float4(0.f, 0.f, 0.f, -1.f), ps, ns;
  if (in) s = *reinterpret_cast<const float4 *>(a.spheres + sidx * 4);
  ps = ns = s;
  if (has_prev) ps = *reinterpret_cast<const float4 *>(a.spheres + (sidx - a.nspheres) * 4);
  if (has_next) ns = *reinterpret_cast<const float4 *>(a.spheres + (sidx + a.nspheres) * 4);
  stash[tid] = s; stash[NT + tid] = ps; stash[2 * NT + tid] = ns;
  const float eta = a.activation_distance[0], w = a.weight[0];
  __syncthreads();
  // ---- phase 1: which obstacles survive the early reject for my sphere
  uint32_t live = 0u;
  const ObsRec *my_recs = recs + (size_t)(b - b_first) * n_rec;
  {
    const f3 center = make_f3(s.x, s.y, s.z);
    if (in && s.w >= 0.0f) {
      const float r_adj = s.w + eta;
      float half_w_prev = 0.0f, half_w_next = 0.0f;
      if (SWEEP > 0) {
        if (has_prev) { const f3 dd = make_f3(ps.x, ps.y, ps.z) - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
        if (has_next) { const f3 dd = make_f3(ns.x, ns.y, ns.z) - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
      }
      const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
      const float thr_c = r_adj + reach, thr2_c = thr_c * thr_c * 1.00001f;
      for (int o = 0; o < n_rec; o++) {
        const ObsRec rec = my_recs[o];
        if (rec.meta.x == 0.0f) continue;
        const f3 lc = to_local(rec, center);
        const bool vox = (KINDS & 2) && (!(KINDS & 1) || o >= a.sc.max_cuboids);
        const bool rej = vox ? obstacle_early_reject<true>(a.sc, rec, lc, r_adj, reach, thr2_c)
                             : obstacle_early_reject<false>(a.sc, rec, lc, r_adj, reach, thr2_c);
        if (!rej) live |= 1u << o;
      }
    }
  }
  // ---- exclusive prefix of the item counts over the workgroup (wave prefix by DPP-free shuffles, then 4 wave totals)
  const int cnt = __builtin_popcount(live);
  int inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(inc, off, 64);
    if (lane64 >= off) inc += v;
  }
  if (lane64 == 63) prefix[NT + wave] = inc;
  __syncthreads();
  int base = 0;
  for (int wv = 0; wv < wave; wv++) base += prefix[NT + wv];
  const int n_items = prefix[NT] + prefix[NT + 1] + prefix[NT + 2] + prefix[NT + 3];
  const int my_first = base + inc - cnt;
  {
    uint32_t m = live;
    int at = my_first;
    while (m) {
      const int o = __ffs((int)m) - 1;
      m &= m - 1;
      queue[at++] = (uint16_t)(tid | (o << 8));
    }
  }
  float dsum = 0.0f;
  f3 gsum = make_f3(0.f, 0.f, 0.f);
  // ---- phase 2: dense evaluation, 256 items per round; owners add their items of the round in queue order
  for (int q0 = 0; q0 < n_items; q0 += NT) {
    __syncthreads();  // queue complete (first round) / results of the previous round consumed
    const int q = q0 + tid;
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < n_items) {
      const unsigned e = queue[q];
      const int owner = (int)(e & 255u), o = (int)(e >> 8);
      const long osidx = sidx0 + owner;
      const int ob = (int)(osidx / hs);
      const int oh = (int)((osidx - (long)ob * hs) / a.nspheres);
      const int env = a.use_multi_env ? a.env_query_idx[ob] : 0;
      const float4 cs = stash[owner], cp = stash[NT + owner], cn = stash[2 * NT + owner];
      const ObsRec rec = recs[(size_t)(ob - b_first) * n_rec + o];
      const f3 center = make_f3(cs.x, cs.y, cs.z), pp = make_f3(cp.x, cp.y, cp.z), np = make_f3(cn.x, cn.y, cn.z);
      const bool hp = need_nb && oh > 0, hn = need_nb && oh < a.horizon - 1;
      const float r_adj = cs.w + eta;
      float half_w_prev = 0.0f, half_w_next = 0.0f;
      if (SWEEP > 0) {
        if (hp) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
        if (hn) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
      }
      const f3 lc = to_local(rec, center);
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      const bool vox = (KINDS & 2) && (!(KINDS & 1) || o >= a.sc.max_cuboids);
      if (vox)
        obstacle_contribution<true, SWEEP>(a.sc, rec, env * a.sc.max_voxel_grids + o - a.sc.max_cuboids, lc, hp, hn, pp, np, r_adj, eta,
                                           half_w_prev, half_w_next, cost_sum, grad_local);
      else
        obstacle_contribution<false, SWEEP, (KINDS & 4) != 0>(a.sc, rec, env * a.sc.max_cuboids + o, lc, hp, hn, pp, np, r_adj, eta, half_w_prev,
                                            half_w_next, cost_sum, grad_local);
      if (cost_sum > 0.0f) {
        const f3 gw = to_world_vector(rec, grad_local);
        r4 = make_float4(w * gw.x, w * gw.y, w * gw.z, w * cost_sum);
      }
    }
    res[tid] = r4;
    __syncthreads();
    const int lo = my_first > q0 ? my_first : q0, hi = (my_first + cnt) < (q0 + NT) ? (my_first + cnt) : (q0 + NT);
    for (int k = lo; k < hi; k++) {
      const float4 v = res[k - q0];
      if (v.w > 0.0f) { dsum += v.w; gsum = gsum + make_f3(v.x, v.y, v.z); }
    }
  }
  if (!in) return;
  if (a.enable_speed_metric && has_prev && has_next && dsum > 0.0f)
    speed_metric_apply(make_f3(s.x, s.y, s.z), make_f3(ps.x, ps.y, ps.z), make_f3(ns.x, ns.y, ns.z), a.speed_dt[0], dsum, gsum);
  a.distance[sidx] = dsum;
  reinterpret_cast<float4 *>(a.gradient)[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_sphere_obstacle_collision(
    float *distance, float *gradient, const float *spheres, const curobo_hip_scene *scene,
    const float *weight, const float *activation_distance, const int32_t *env_query_idx,
    int batch_size, int horizon, int num_spheres, int use_multi_env, int sweep_steps,
    int enable_speed_metric, const float *speed_dt, curobo_hip_stream_t stream) {
  const char *what = "sphere_obstacle_collision";
  CUROBO_REQUIRE(scene != nullptr, "%s: scene is NULL", what);
  CUROBO_REQUIRE(sweep_steps == 0 || sweep_steps == 3, "%s: sweep_steps must be 0 or 3 (got %d)", what, sweep_steps);
  CUROBO_REQUIRE(!enable_speed_metric || speed_dt, "%s: speed metric needs speed_dt", what);
  CUROBO_REQUIRE(!use_multi_env || env_query_idx, "%s: use_multi_env needs env_query_idx", what);
  CUROBO_REQUIRE(scene->max_cuboids >= 0 && scene->max_voxel_grids >= 0, "%s: negative obstacle capacity", what);
  const long total = (long)batch_size * horizon * num_spheres;
  if (total == 0) return CUROBO_HIP_OK;
  SceneArgs a{};
  a.distance = distance; a.gradient = gradient; a.spheres = spheres; a.sc = *scene;
  a.weight = weight; a.activation_distance = activation_distance; a.env_query_idx = env_query_idx;
  a.speed_dt = speed_dt; a.batch = batch_size; a.horizon = horizon; a.nspheres = num_spheres;
  a.use_multi_env = use_multi_env; a.sweep_steps = sweep_steps; a.enable_speed_metric = enable_speed_metric;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)ceil_div_l(total, 256);
  // obstacle records of every batch row a 256-sphere workgroup can touch, staged in LDS
  const long hs = (long)horizon * num_spheres;
  const long slots = (256 + hs - 1) / hs + 1;
  const size_t lds = (size_t)slots * (scene->max_cuboids + scene->max_voxel_grids) * sizeof(ObsRec);
  const bool staged = lds > 0 && lds <= 32 * 1024;
  const int kinds = (scene->max_cuboids > 0 ? 1 : 0) | (scene->max_voxel_grids > 0 ? 2 : 0) |
                    ((scene->max_cuboids > 0 && scene->cuboid_has_primitives) ? 4 : 0);
#define CUROBO_SCENE_LAUNCH(SW, ST, KD) \
  hipLaunchKernelGGL((scene_collision_kernel<SW, ST, KD>), dim3(blocks), dim3(256), (ST) ? lds : 0, st, a)
#define CUROBO_SCENE_KINDS(SW, ST)                   \
  do {                                               \
    if (kinds == 1) CUROBO_SCENE_LAUNCH(SW, ST, 1);  \
    else if (kinds == 2) CUROBO_SCENE_LAUNCH(SW, ST, 2); \
    else if (kinds == 5) CUROBO_SCENE_LAUNCH(SW, ST, 5); \
    else if (kinds == 7) CUROBO_SCENE_LAUNCH(SW, ST, 7); \
    else CUROBO_SCENE_LAUNCH(SW, ST, 3);             \
  } while (0)
  const int n_rec = scene->max_cuboids + scene->max_voxel_grids;
  // (the queue holds at most one item per sphere and obstacle record: sized by the scene, not by the 32-record limit --
  // LDS per workgroup is occupancy)
  const size_t lds_packed = lds + (size_t)(3 + 1) * 256 * 16 + (256 + 8) * 4 + (((size_t)256 * n_rec * 2 + 15) & ~(size_t)15);
  static const bool no_packed = getenv("CUROBO_HIP_SCENE_UNPACKED") != nullptr;  // development knob: the in-lane obstacle loop
  if (staged && (kinds & 2) && n_rec <= 32 && lds_packed <= 64 * 1024 && !no_packed) {  // ESDF grids: see the kernel's header
#define CUROBO_SCENE_PACKED(SW)                                                                                              \
  do {                                                                                                                       \
    if (kinds == 1) hipLaunchKernelGGL((scene_collision_packed_kernel<SW, 1>), dim3(blocks), dim3(256), lds_packed, st, a);      \
    else if (kinds == 2) hipLaunchKernelGGL((scene_collision_packed_kernel<SW, 2>), dim3(blocks), dim3(256), lds_packed, st, a); \
    else if (kinds == 7) hipLaunchKernelGGL((scene_collision_packed_kernel<SW, 7>), dim3(blocks), dim3(256), lds_packed, st, a); \
    else hipLaunchKernelGGL((scene_collision_packed_kernel<SW, 3>), dim3(blocks), dim3(256), lds_packed, st, a);                 \
  } while (0)
    if (sweep_steps == 0) CUROBO_SCENE_PACKED(0); else CUROBO_SCENE_PACKED(3);
#undef CUROBO_SCENE_PACKED
  } else if (sweep_steps == 0) {
    if (staged) CUROBO_SCENE_KINDS(0, true); else CUROBO_SCENE_KINDS(0, false);
  } else {
    if (staged) CUROBO_SCENE_KINDS(3, true); else CUROBO_SCENE_KINDS(3, false);
  }
#undef CUROBO_SCENE_KINDS
#undef CUROBO_SCENE_LAUNCH
  return check_launch(what, st);
}
