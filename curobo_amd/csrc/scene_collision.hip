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
  const int kinds = (scene->max_cuboids > 0 ? 1 : 0) | (scene->max_voxel_grids > 0 ? 2 : 0);
#define CUROBO_SCENE_LAUNCH(SW, ST, KD) \
  hipLaunchKernelGGL((scene_collision_kernel<SW, ST, KD>), dim3(blocks), dim3(256), (ST) ? lds : 0, st, a)
#define CUROBO_SCENE_KINDS(SW, ST)                   \
  do {                                               \
    if (kinds == 1) CUROBO_SCENE_LAUNCH(SW, ST, 1);  \
    else if (kinds == 2) CUROBO_SCENE_LAUNCH(SW, ST, 2); \
    else CUROBO_SCENE_LAUNCH(SW, ST, 3);             \
  } while (0)
  if (sweep_steps == 0) {
    if (staged) CUROBO_SCENE_KINDS(0, true); else CUROBO_SCENE_KINDS(0, false);
  } else {
    if (staged) CUROBO_SCENE_KINDS(3, true); else CUROBO_SCENE_KINDS(3, false);
  }
#undef CUROBO_SCENE_KINDS
#undef CUROBO_SCENE_LAUNCH
  return check_launch(what, st);
}
