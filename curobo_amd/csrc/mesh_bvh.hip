// mesh_bvh.hip -- triangle-mesh obstacles on the device: a linear BVH, the closest-point / sign query, sphere-vs-mesh
// collision (discrete, swept, speed metric) and the ESDF bake through the BVH.
//
// Reference: curobo/_src/geom/data/data_mesh.py:555-700 -- per query sphere wp.mesh_query_point(mesh, point, max_distance)
// -> closest point on the surface -> signed distance (negative inside) and the unit vector (point - closest point) as the
// local gradient; max_distance = max(half the diagonal of the mesh's bounding box, the query distance), a query that finds
// nothing within it returns (max_distance, 0).  The BVH and the sign are NVIDIA Warp's (warp-lang, not in the reference
// tree, no ROCm backend): what is restated here is the published contract of mesh_query_point, not its code.
//
// MI355X design.  A linear BVH in heap layout: the triangles are sorted by the Morton code of their centroids (codes from
// a kernel here, the sort is the caller's -- torch.sort, plumbing), `leaf_size` consecutive triangles form a leaf, the
// leaf count is padded to a power of two and node k has the children 2k and 2k + 1: no pointers, no build-time atomics,
// the boxes of a level are one launch.  Triangles are stored in sorted order as (a, b - a, c - a) float4 triples so that
// a leaf is one contiguous run.  The query is a per-lane stack traversal (nearer child first, prune by the best squared
// distance so far); the sign is the parity of ray crossings, majority of three rays through the same BVH (closed meshes;
// the oracle uses the generalised winding number instead: two independent methods that must agree).
#include "common.hpp"

#include <hip/hip_fp16.h>

namespace curobo_hip {


constexpr int kMeshStack = 64;

struct TriRec {  // 48 bytes
  float4 a, ab, ac;
};

__device__ __forceinline__ float box_dist2(const float4 lo, const float4 hi, f3 p) {
  const float dx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.0f), dy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.0f),
              dz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.0f);
  return dx * dx + dy * dy + dz * dz;
}

// closest point of triangle (a, a + ab, a + ac) to p (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ f3 closest_on_triangle(f3 p, f3 a, f3 ab, f3 ac, bool &interior) {
  interior = false;
  const f3 ap = p - a;
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;
  const f3 b = a + ab, bp = p - b;
  const float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) return a + (d1 / (d1 - d3)) * ab;
  const f3 c = a + ac, cp = p - c;
  const float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return c;
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) return a + (d2 / (d2 - d6)) * ac;
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) return b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
  const float den = 1.0f / (va + vb + vc);
  interior = true;  // the face region: the closest point is the projection on the triangle's plane
  return a + (vb * den) * ab + (vc * den) * ac;
}

// closest surface point within sqrt(best_d2) of p; returns false when there is none
// side: +1 / -1 = p is on the outer / inner side of the FACE its closest point lies in (then that is the sign of the
// signed distance of a closed mesh), 0 = the closest point lies on an edge or a vertex (no verdict: count crossings)
__device__ __forceinline__ bool mesh_closest_point(const curobo_hip_mesh &m, f3 p, float &best_d2, f3 &cp, int &side) {
  side = 0;
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  int stack[kMeshStack];
  int sp = 0;
  bool found = false;
  if (box_dist2(box[2], box[3], p) <= best_d2) stack[sp++] = 1;
  while (sp > 0) {
    const int node = stack[--sp];
    if (box_dist2(box[node * 2], box[node * 2 + 1], p) > best_d2) continue;  // the best may have shrunk since the push
    if (node >= m.n_leaves) {
      const int t0 = (node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
      for (int t = t0; t < t1; t++) {
        const TriRec r = tri[t];
        bool interior;
        const f3 ab = make_f3(r.ab.x, r.ab.y, r.ab.z), ac = make_f3(r.ac.x, r.ac.y, r.ac.z);
        const f3 c = closest_on_triangle(p, make_f3(r.a.x, r.a.y, r.a.z), ab, ac, interior);
        const f3 d = p - c;
        const float d2 = dot(d, d);
        if (d2 <= best_d2) {
          // (a tie between a face and an edge / vertex of a neighbour keeps whichever came last; a face verdict is only
          // trusted when the point is clearly off the plane)
          const float sd = dot(d, cross(ab, ac));
          side = (interior && d2 > 1e-12f && sd != 0.0f) ? (sd > 0.0f ? 1 : -1) : 0;
          if (d2 == best_d2 && found) side = 0;
          best_d2 = d2; cp = c; found = true;
        }
      }
    } else {
      const int c0 = node * 2, c1 = c0 + 1;
      const float d0 = box_dist2(box[c0 * 2], box[c0 * 2 + 1], p), d1 = box_dist2(box[c1 * 2], box[c1 * 2 + 1], p);
      // the farther child first: the nearer one is popped next
      if (d0 <= d1) {
        if (d1 <= best_d2 && sp < kMeshStack) stack[sp++] = c1;
        if (d0 <= best_d2 && sp < kMeshStack) stack[sp++] = c0;
      } else {
        if (d0 <= best_d2 && sp < kMeshStack) stack[sp++] = c0;
        if (d1 <= best_d2 && sp < kMeshStack) stack[sp++] = c1;
      }
    }
  }
  return found;
}

// crossings of the ray p + t d (t > 0) with the surface
__device__ __forceinline__ int mesh_ray_crossings(const curobo_hip_mesh &m, f3 p, f3 d) {
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  const f3 inv = make_f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int stack[kMeshStack];
  int sp = 0, hits = 0;
  stack[sp++] = 1;
  while (sp > 0) {
    const int node = stack[--sp];
    const float4 lo = box[node * 2], hi = box[node * 2 + 1];
    // slab test (an empty padding box has lo > hi: t_enter > t_exit)
    const float tx0 = (lo.x - p.x) * inv.x, tx1 = (hi.x - p.x) * inv.x, ty0 = (lo.y - p.y) * inv.y, ty1 = (hi.y - p.y) * inv.y,
                tz0 = (lo.z - p.z) * inv.z, tz1 = (hi.z - p.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
    const float t_out = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    if (!(t_in <= t_out) || lo.x > hi.x) continue;
    if (node >= m.n_leaves) {
      const int t0 = (node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
      for (int t = t0; t < t1; t++) {  // Moeller-Trumbore
        const TriRec r = tri[t];
        const f3 ab = make_f3(r.ab.x, r.ab.y, r.ab.z), ac = make_f3(r.ac.x, r.ac.y, r.ac.z);
        const f3 pv = cross(d, ac);
        const float det = dot(ab, pv);
        if (fabsf(det) < 1e-20f) continue;
        const float idet = 1.0f / det;
        const f3 tv = p - make_f3(r.a.x, r.a.y, r.a.z);
        const float u = dot(tv, pv) * idet;
        if (u < 0.0f || u > 1.0f) continue;
        const f3 qv = cross(tv, ab);
        const float v = dot(d, qv) * idet;
        if (v < 0.0f || u + v > 1.0f) continue;
        if (dot(ac, qv) * idet > 0.0f) hits++;
      }
    } else if (sp + 2 <= kMeshStack) {
      stack[sp++] = node * 2;
      stack[sp++] = node * 2 + 1;
    }
  }
  return hits;
}

// inside a closed mesh: the parity of surface crossings, majority of three rays in generic directions (a ray that
// grazes an edge or a vertex may count a crossing twice or not at all; three unrelated directions do not all do)
__device__ __forceinline__ bool mesh_inside(const curobo_hip_mesh &m, f3 p) {
  const int a = mesh_ray_crossings(m, p, make_f3(1.0f, 0.0713f, 0.0291f)) & 1;
  const int b = mesh_ray_crossings(m, p, make_f3(-0.0517f, 1.0f, 0.0839f)) & 1;
  if (a == b) return a != 0;
  return (mesh_ray_crossings(m, p, make_f3(0.0331f, -0.0617f, -1.0f)) & 1) != 0;
}

// data_mesh.py:630-700 compute_local_sdf_with_grad: signed distance (negative inside) and the local gradient
// (p - closest) / |p - closest| -- as the reference returns it, whatever side p is on.  No surface within max_distance:
// (max_distance, 0).
__device__ __forceinline__ float mesh_sdf_with_grad(const curobo_hip_mesh &m, f3 lp, float max_distance, f3 &g) {
  g = make_f3(0.f, 0.f, 0.f);
  float d2 = max_distance * max_distance;
  f3 cp = lp;
  int side;
  if (!mesh_closest_point(m, lp, d2, cp, side)) return max_distance;
  const float d = sqrtf(d2);
  const f3 delta = lp - cp;
  if (d > 1e-6f) g = (1.0f / d) * delta;
  // the face the closest point lies in says which side the point is on; on an edge or a vertex the crossings are counted
  const bool inside = side != 0 ? side < 0 : mesh_inside(m, lp);
  return inside ? -d : d;
}

// ------------------------------------------------------------------------------------------------ build
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void __launch_bounds__(256) mesh_morton_kernel(int64_t *codes, const float *vertices, const int32_t *faces, int n_tri,
                                                          float lx, float ly, float lz, float sx, float sy, float sz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tri) return;
  float c[3] = {0.f, 0.f, 0.f};
  for (int v = 0; v < 3; v++)
    for (int ax = 0; ax < 3; ax++) c[ax] += vertices[(size_t)faces[(size_t)t * 3 + v] * 3 + ax] * (1.0f / 3.0f);
  const uint32_t x = (uint32_t)fminf(fmaxf((c[0] - lx) * sx, 0.0f), 1023.0f), y = (uint32_t)fminf(fmaxf((c[1] - ly) * sy, 0.0f), 1023.0f),
                 z = (uint32_t)fminf(fmaxf((c[2] - lz) * sz, 0.0f), 1023.0f);
  // ties keep the triangle order (the index in the low bits makes the keys unique: any sort gives the same permutation)
  codes[t] = ((int64_t)((spread3(x) << 2) | (spread3(y) << 1) | spread3(z)) << 32) | (int64_t)t;
}

__global__ void __launch_bounds__(256) mesh_leaves_kernel(float *out_tri, float *out_box, const float *vertices, const int32_t *faces,
                                                          const int64_t *sorted_codes, int n_tri, int n_leaves, int leaf_size) {
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n_leaves) return;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int t = leaf * leaf_size; t < min((leaf + 1) * leaf_size, n_tri); t++) {
    const int src = (int)(sorted_codes[t] & 0xffffffffll);
    float v[3][3];
    for (int k = 0; k < 3; k++)
      for (int ax = 0; ax < 3; ax++) {
        v[k][ax] = vertices[(size_t)faces[(size_t)src * 3 + k] * 3 + ax];
        lo[ax] = fminf(lo[ax], v[k][ax]);
        hi[ax] = fmaxf(hi[ax], v[k][ax]);
      }
    float4 *o = reinterpret_cast<float4 *>(out_tri) + (size_t)t * 3;
    o[0] = make_float4(v[0][0], v[0][1], v[0][2], 0.f);
    o[1] = make_float4(v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2], 0.f);
    o[2] = make_float4(v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2], 0.f);
  }
  float4 *b = reinterpret_cast<float4 *>(out_box) + (size_t)(n_leaves + leaf) * 2;
  b[0] = make_float4(lo[0], lo[1], lo[2], 0.f);
  b[1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

__global__ void __launch_bounds__(256) mesh_level_kernel(float *box, int first, int count) {  // nodes first .. first + count - 1
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int node = first + i;
  float4 *b = reinterpret_cast<float4 *>(box);
  const float4 l0 = b[(size_t)node * 4], h0 = b[(size_t)node * 4 + 1], l1 = b[(size_t)node * 4 + 2], h1 = b[(size_t)node * 4 + 3];
  b[(size_t)node * 2] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.f);
  b[(size_t)node * 2 + 1] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
}

// ------------------------------------------------------------------------------------------------ queries
__global__ void __launch_bounds__(256) mesh_query_kernel(float *out_sdf, float *out_grad, const float *points, const curobo_hip_mesh m,
                                                         float max_distance, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3 g;
  const float sdf = mesh_sdf_with_grad(m, make_f3(points[i * 3], points[i * 3 + 1], points[i * 3 + 2]), max_distance, g);
  out_sdf[i] = sdf;
  if (out_grad) { out_grad[i * 3] = g.x; out_grad[i * 3 + 1] = g.y; out_grad[i * 3 + 2] = g.z; }
}

struct MeshBakeBvhArgs {
  __half *out;
  curobo_hip_mesh m;
  int nx, ny, nz;
  float voxel_size, max_distance;
  float g2m[12];
};

__global__ void __launch_bounds__(256) mesh_esdf_bake_bvh_kernel(const MeshBakeBvhArgs a) {
  const long n_vox = (long)a.nx * a.ny * a.nz;
  const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vox) return;
  const int iz = (int)(v % a.nz), iy = (int)((v / a.nz) % a.ny), ix = (int)(v / ((long)a.nz * a.ny));
  const f3 g = make_f3(((float)ix + 0.5f - 0.5f * (float)a.nx) * a.voxel_size, ((float)iy + 0.5f - 0.5f * (float)a.ny) * a.voxel_size,
                       ((float)iz + 0.5f - 0.5f * (float)a.nz) * a.voxel_size);
  const f3 p = make_f3(a.g2m[0] * g.x + a.g2m[1] * g.y + a.g2m[2] * g.z + a.g2m[3], a.g2m[4] * g.x + a.g2m[5] * g.y + a.g2m[6] * g.z + a.g2m[7],
                       a.g2m[8] * g.x + a.g2m[9] * g.y + a.g2m[10] * g.z + a.g2m[11]);
  // the field is clamped to +-max_distance anyway: nothing farther needs a closest point, only a side
  float d2 = a.max_distance * a.max_distance;
  f3 cp = p;
  int side;
  const bool found = mesh_closest_point(a.m, p, d2, cp, side);
  const float d = found ? sqrtf(d2) : a.max_distance;
  const bool inside = (found && side != 0) ? side < 0 : mesh_inside(a.m, p);
  a.out[v] = __float2half(inside ? -d : d);
}

// ------------------------------------------------------------------------------------------------ sphere vs meshes
struct MeshCollArgs {
  float *distance, *gradient;
  const float *spheres;
  curobo_hip_mesh_set set;
  const float *weight, *eta, *speed_dt;
  const int32_t *env_query_idx;
  int batch, horizon, nspheres, use_multi_env, enable_speed_metric, accumulate;
};

__device__ __forceinline__ void activation_m(float dist, float eta, float &cost, float &gscale) {  // wp_collision_common.py:11-38
  if (dist > eta) { cost = dist - 0.5f * eta; gscale = 1.0f; }
  else { cost = 0.5f * dist * dist / eta; gscale = dist / eta; }
}

__device__ __forceinline__ float mesh_point_terms(const curobo_hip_mesh &m, f3 lp, float max_distance, float r_adj, float eta,
                                                  float &c, float &gs, f3 &g, int gradient_mode) {
  const float sdf = mesh_sdf_with_grad(m, lp, max_distance, g);
  if (gradient_mode == 1 && sdf > 0.0f) g = -1.0f * g;
  const float pen = -sdf + r_adj;
  c = 0.0f; gs = 0.0f;
  if (pen > 0.0f) activation_m(pen, eta, c, gs);
  return pen;
}
__device__ __forceinline__ float mesh_eval_point(const curobo_hip_mesh &m, f3 lp, float max_distance, float r_adj, float eta,
                                                 float &cost_sum, f3 &grad_sum, int gradient_mode) {
  f3 g;
  float c, gs;
  const float pen = mesh_point_terms(m, lp, max_distance, r_adj, eta, c, gs, g, gradient_mode);
  if (pen > 0.0f) {
    cost_sum += c;
    grad_sum = grad_sum + gs * g;
  }
  return pen;
}

__device__ __forceinline__ f3 quat_rot(float qw, float qx, float qy, float qz, f3 v) {  // warp quat_rotate
  const f3 q = make_f3(qx, qy, qz);
  const float c = 2.0f * qw * qw - 1.0f, d = 2.0f * dot(q, v);
  const f3 cr = cross(q, v);
  return make_f3(v.x * c + q.x * d + cr.x * 2.0f * qw, v.y * c + q.y * d + cr.y * 2.0f * qw, v.z * c + q.z * d + cr.z * 2.0f * qw);
}

template <int SWEEP>
__global__ void __launch_bounds__(256) sphere_mesh_collision_kernel(const MeshCollArgs a) {
  const long total = (long)a.batch * a.horizon * a.nspheres;
  const long sidx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= total) return;
  const int b = (int)(sidx / ((long)a.horizon * a.nspheres));
  const int h = (int)((sidx - (long)b * a.horizon * a.nspheres) / a.nspheres);
  const int env = a.use_multi_env ? a.env_query_idx[b] : 0;
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const float4 s = sph[sidx];
  const float w = a.weight[0], eta = a.eta[0];
  float dsum = 0.0f;
  f3 gsum = make_f3(0.f, 0.f, 0.f);
  const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;  // neighbours feed the sweep and the speed metric
  const bool nb_prev = need_nb && h > 0, nb_next = need_nb && h < a.horizon - 1;
  const bool has_prev = SWEEP > 0 && nb_prev, has_next = SWEEP > 0 && nb_next;
  const float4 ps = nb_prev ? sph[sidx - a.nspheres] : s, ns = nb_next ? sph[sidx + a.nspheres] : s;
  const f3 center = make_f3(s.x, s.y, s.z), pp = make_f3(ps.x, ps.y, ps.z), np = make_f3(ns.x, ns.y, ns.z);
  if (s.w >= 0.0f) {
    const float r_adj = s.w + eta;
    float half_w_prev = 0.0f, half_w_next = 0.0f;
    if (SWEEP > 0) {
      if (has_prev) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (has_next) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
    const curobo_hip_mesh_set &ms = a.set;
    const int count = ms.count[env];
    for (int o = 0; o < ms.max_n; o++) {
      const int flat = env * ms.max_n + o;
      if (o >= count || ms.enable[flat] != 1) continue;  // is_obs_enabled (data_mesh.py:555-575)
      const curobo_hip_mesh m = ms.meshes[ms.mesh_id[flat]];
      const float *ip = ms.inv_pose + (size_t)flat * 8;  // x y z qw qx qy qz pad: world -> mesh frame
      const f3 t = make_f3(ip[0], ip[1], ip[2]);
      const float qw = ip[3], qx = ip[4], qy = ip[5], qz = ip[6];
      const float *dm = ms.dims + (size_t)flat * 4;
      // max_distance = max(half the bounding-box diagonal, the query distance) (data_mesh.py:660-668)
      const float max_distance = fmaxf(0.5f * sqrtf(dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2]), r_adj);
      const f3 lc = quat_rot(qw, qx, qy, qz, center) + t;
      // Early reject (result preserving): the surface lies inside the mesh's bounding box (root of the tree), so the
      // signed distance of a point outside the box is at least its distance to the box; when that exceeds
      // r_adj + the half sweep length (+ rounding) neither the centre nor any sweep sample can penetrate.
      {
        const float *rb = m.node_box + 8;
        const float ex = fmaxf(fmaxf(rb[0] - lc.x, lc.x - rb[4]), 0.0f), ey = fmaxf(fmaxf(rb[1] - lc.y, lc.y - rb[5]), 0.0f),
                    ez = fmaxf(fmaxf(rb[2] - lc.z, lc.z - rb[6]), 0.0f);
        const float thr = r_adj + reach;
        if (ex * ex + ey * ey + ez * ez > thr * thr * 1.00001f) continue;
      }
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      f3 g_c;
      float c_c, gs_c;
      const float pen_c = mesh_point_terms(m, lc, max_distance, r_adj, eta, c_c, gs_c, g_c, ms.gradient_mode);
      if (pen_c > 0.0f) { cost_sum += c_c; grad_local = grad_local + gs_c * g_c; }
      const float sdf_c = r_adj - pen_c;
      if (SWEEP > 0) {  // wp_sweep_collision_kernel.py:176-254
#pragma unroll 1
        for (int dir = 0; dir < 2; dir++) {
          if (!(dir == 0 ? has_prev : has_next)) continue;
          // sweep culling (result preserving, as for cuboids: scene_device.hpp): every sample lies within the half
          // segment length of the centre and the signed distance is 1-Lipschitz, so a centre that is clear by more
          // than that cannot have a penetrating sample (no bound when the centre found no surface within max_distance)
          if (sdf_c < max_distance && -pen_c > (dir == 0 ? half_w_prev : half_w_next) * 1.0001f + 2e-6f + 1e-6f * max_distance) continue;
          const f3 ln = quat_rot(qw, qx, qy, qz, dir == 0 ? pp : np) + t;
          const f3 dd = ln - lc;
          const float half_dist = sqrtf(dot(dd, dd)) * 0.5f;
          const float inv_half = 1.0f / fmaxf(half_dist, 0.001f);
          float jump = 0.0f;
          if (jump >= half_dist) continue;
          // k = 0 of the reference's loop samples t = 1, the centre itself: its terms are added again, not walked again
          if (pen_c > 0.0f) { cost_sum += c_c; grad_local = grad_local + gs_c * g_c; jump += pen_c; }
          else if (-pen_c >= 1000.0f) jump += r_adj;
          else jump += fmaxf(-pen_c, r_adj);
          for (int k = 1; k < SWEEP; k++) {
            if (jump >= half_dist) break;
            const float tt = 1.0f - 0.5f * jump * inv_half;
            const f3 lp = tt * lc + (1.0f - tt) * ln;
            const float p2 = mesh_eval_point(m, lp, max_distance, r_adj, eta, cost_sum, grad_local, ms.gradient_mode);
            if (p2 > 0.0f) jump += p2;
            else if (-p2 >= 1000.0f) jump += r_adj;
            else jump += fmaxf(-p2, r_adj);
          }
        }
      }
      if (cost_sum > 0.0f) {
        const f3 gw = quat_rot(qw, -qx, -qy, -qz, grad_local);  // transform_vector(transform_inverse(inv_t), .)
        dsum += w * cost_sum;
        gsum = gsum + w * gw;
      }
    }
  }
  float4 *grad = reinterpret_cast<float4 *>(a.gradient);
  if (a.accumulate) {
    // the other obstacle kinds were written (and, when on, speed scaled) by the scene launch before this one: the speed
    // metric is linear in (cost, gradient), so scaling this kind's share on its own and adding gives the same sums as the
    // reference's one scaling pass over all kinds (wp_autograd.py:213-231) -- except for its `cost > 0` guard, which the
    // sum passes whenever a share does
    if (dsum > 0.0f) {
      if (a.enable_speed_metric && h > 0 && h < a.horizon - 1) {
        float dt = a.speed_dt[0];
        if (dt < 1e-6f) dt = 1e-6f;
        const f3 vel = (0.5f / dt) * (np - pp);
        const float sv = sqrtf(dot(vel, vel));
        if (sv >= 1e-3f) {
          const f3 acc = (1.0f / (dt * dt)) * (pp + np - 2.0f * center);
          const f3 nv = make_f3(vel.x / sv, vel.y / sv, vel.z / sv);
          const float sv2 = sv * sv;
          const f3 curv = make_f3(acc.x / sv2, acc.y / sv2, acc.z / sv2);
          const f3 og = gsum - dot(nv, gsum) * nv;
          const f3 oc = curv - dot(nv, curv) * nv;
          gsum = sv * (og - dsum * oc);
          dsum = sv * dsum;
        }
      }
      a.distance[sidx] += dsum;
      const float4 g0 = grad[sidx];
      grad[sidx] = make_float4(g0.x + gsum.x, g0.y + gsum.y, g0.z + gsum.z, g0.w);
    }
    return;
  }
  if (a.enable_speed_metric && h > 0 && h < a.horizon - 1 && dsum > 0.0f) {
    float dt = a.speed_dt[0];
    if (dt < 1e-6f) dt = 1e-6f;
    const f3 vel = (0.5f / dt) * (np - pp);
    const float sv = sqrtf(dot(vel, vel));
    if (sv >= 1e-3f) {
      const f3 acc = (1.0f / (dt * dt)) * (pp + np - 2.0f * center);
      const f3 nv = make_f3(vel.x / sv, vel.y / sv, vel.z / sv);
      const float sv2 = sv * sv;
      const f3 curv = make_f3(acc.x / sv2, acc.y / sv2, acc.z / sv2);
      const f3 og = gsum - dot(nv, gsum) * nv;
      const f3 oc = curv - dot(nv, curv) * nv;
      gsum = sv * (og - dsum * oc);
      dsum = sv * dsum;
    }
  }
  a.distance[sidx] = dsum;
  grad[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_mesh_morton_codes(int64_t *out_codes, const float *vertices, const int32_t *faces, int n_faces,
                                               const float *bounds_lo_hi_host, curobo_hip_stream_t stream) {
  const char *what = "mesh_morton_codes";
  CUROBO_REQUIRE(out_codes && vertices && faces && bounds_lo_hi_host && n_faces > 0, "%s: bad arguments", what);
  const float *b = bounds_lo_hi_host;
  float sc[3];
  for (int i = 0; i < 3; i++) sc[i] = 1023.0f / fmaxf(b[3 + i] - b[i], 1e-12f);
  hipLaunchKernelGGL(mesh_morton_kernel, dim3((unsigned)ceil_div(n_faces, 256)), dim3(256), 0, (hipStream_t)stream, out_codes, vertices,
                     faces, n_faces, b[0], b[1], b[2], sc[0], sc[1], sc[2]);
  return check_launch(what, (hipStream_t)stream);
}

CUROBO_EXPORT int curobo_hip_mesh_bvh_build(float *out_tri, float *out_node_box, const float *vertices, const int32_t *faces,
                                            const int64_t *sorted_codes, int n_faces, int n_leaves, int leaf_size,
                                            curobo_hip_stream_t stream) {
  const char *what = "mesh_bvh_build";
  CUROBO_REQUIRE(out_tri && out_node_box && vertices && faces && sorted_codes, "%s: NULL argument", what);
  CUROBO_REQUIRE(n_faces > 0 && leaf_size >= 1 && n_leaves >= 1 && (n_leaves & (n_leaves - 1)) == 0 &&
                     (long)n_leaves * leaf_size >= n_faces && n_leaves <= (1 << 24),
                 "%s: n_leaves (%d) must be a power of two with n_leaves * leaf_size (%d) >= n_faces (%d)", what, n_leaves, leaf_size, n_faces);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mesh_leaves_kernel, dim3((unsigned)ceil_div(n_leaves, 256)), dim3(256), 0, st, out_tri, out_node_box, vertices, faces,
                     sorted_codes, n_faces, n_leaves, leaf_size);
  for (int count = n_leaves / 2; count >= 1; count /= 2)  // level of `count` nodes: first node = count
    hipLaunchKernelGGL(mesh_level_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, st, out_node_box, count, count);
  return check_launch(what, st);
}

static int check_mesh(const curobo_hip_mesh *m, const char *what) {
  CUROBO_REQUIRE(m && m->tri && m->node_box && m->n_tri > 0 && m->n_leaves >= 1 && m->leaf_size >= 1, "%s: incomplete mesh", what);
  // stack: the traversal pushes at most two nodes per level
  CUROBO_REQUIRE(m->n_leaves <= (1 << 24), "%s: mesh too large", what);
  return CUROBO_HIP_OK;
}

CUROBO_EXPORT int curobo_hip_mesh_query(float *out_sdf, float *out_grad, const float *points, const curobo_hip_mesh *mesh,
                                        float max_distance, int n_points, curobo_hip_stream_t stream) {
  const char *what = "mesh_query";
  CUROBO_REQUIRE(out_sdf && points && n_points >= 0 && max_distance > 0.0f, "%s: bad arguments", what);
  if (int rc = check_mesh(mesh, what)) return rc;
  if (n_points == 0) return CUROBO_HIP_OK;
  hipLaunchKernelGGL(mesh_query_kernel, dim3((unsigned)ceil_div(n_points, 256)), dim3(256), 0, (hipStream_t)stream, out_sdf, out_grad,
                     points, *mesh, max_distance, n_points);
  return check_launch(what, (hipStream_t)stream);
}

CUROBO_EXPORT int curobo_hip_mesh_esdf_bake_bvh(uint16_t *out_esdf_fp16, const curobo_hip_mesh *mesh, int nx, int ny, int nz,
                                                float voxel_size, float max_distance, const float *grid_to_mesh_3x4_host,
                                                curobo_hip_stream_t stream) {
  const char *what = "mesh_esdf_bake_bvh";
  CUROBO_REQUIRE(out_esdf_fp16 && grid_to_mesh_3x4_host && nx > 0 && ny > 0 && nz > 0 && voxel_size > 0.0f && max_distance > 0.0f,
                 "%s: bad arguments", what);
  if (int rc = check_mesh(mesh, what)) return rc;
  MeshBakeBvhArgs a{};
  a.out = reinterpret_cast<__half *>(out_esdf_fp16); a.m = *mesh; a.nx = nx; a.ny = ny; a.nz = nz;
  a.voxel_size = voxel_size; a.max_distance = max_distance;
  for (int i = 0; i < 12; i++) a.g2m[i] = grid_to_mesh_3x4_host[i];
  const long n_vox = (long)nx * ny * nz;
  hipLaunchKernelGGL(mesh_esdf_bake_bvh_kernel, dim3((unsigned)ceil_div_l(n_vox, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch(what, (hipStream_t)stream);
}

CUROBO_EXPORT int curobo_hip_sphere_mesh_collision(
    float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres, int use_multi_env,
    int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate, curobo_hip_stream_t stream) {
  const char *what = "sphere_mesh_collision";
  CUROBO_REQUIRE(distance && gradient && spheres && meshes && weight && activation_distance, "%s: NULL argument", what);
  CUROBO_REQUIRE(sweep_steps == 0 || sweep_steps == 3, "%s: sweep_steps must be 0 or 3 (reference SWEEP_STEPS)", what);
  CUROBO_REQUIRE(!use_multi_env || env_query_idx, "%s: use_multi_env needs env_query_idx", what);
  CUROBO_REQUIRE(!enable_speed_metric || speed_dt, "%s: speed metric needs speed_dt", what);
  CUROBO_REQUIRE(meshes->max_n == 0 || (meshes->meshes && meshes->mesh_id && meshes->inv_pose && meshes->dims && meshes->enable && meshes->count),
                 "%s: incomplete mesh set", what);
  const long total = (long)batch_size * horizon * num_spheres;
  if (total == 0 || meshes->max_n == 0) return CUROBO_HIP_OK;
  MeshCollArgs a{};
  a.distance = distance; a.gradient = gradient; a.spheres = spheres; a.set = *meshes; a.weight = weight; a.eta = activation_distance;
  a.speed_dt = speed_dt; a.env_query_idx = env_query_idx; a.batch = batch_size; a.horizon = horizon; a.nspheres = num_spheres;
  a.use_multi_env = use_multi_env; a.enable_speed_metric = enable_speed_metric; a.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)ceil_div_l(total, 256)), block(256);
  if (sweep_steps > 0) hipLaunchKernelGGL((sphere_mesh_collision_kernel<3>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((sphere_mesh_collision_kernel<0>), grid, block, 0, st, a);
  return check_launch(what, st);
}
