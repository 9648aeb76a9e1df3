// scene_device.hpp -- device functions of the sphere-vs-scene collision cost, shared by
// scene_collision.hip and the fused rollout kernels (rollout_fused.hip).
// Reference (NVIDIA Warp): geom/collision/wp_collision_kernel.py:70-166,
// wp_sweep_collision_kernel.py:83-260, wp_speed_metric.py:10-93, wp_collision_common.py:11-96,
// geom/data/data_cuboid.py:461-628, geom/data/data_voxel.py:709-1215, geom/data/helper_pose.py.
#pragma once
#include "common.hpp"

#include <hip/hip_fp16.h>

namespace curobo_hip {

// warp-lang quat_rotate(q, v) = v (2w^2 - 1) + 2w (q x v) + 2 q (q . v), written out as the 3x3
// matrix it is: R = (2w^2 - 1) I + 2w [q]x + 2 q q^T (also for non-unit q).  The obstacle records
// carry R | t of the INVERSE obstacle pose (helper_pose.py:28-90: [x y z qw qx qy qz pad]), built
// once per workgroup, so a sphere-obstacle test starts with 9 FMAs instead of the quaternion form.
struct ObsRec {
  float4 r0, r1, r2;  // rows of R, .w = translation component: local = R world + t
  float4 shape;       // cuboid store: HALF extents xyz of the (bounding) box, .w = primitive tag (0 cuboid, 1 sphere,
                      // 2 capsule, 3 cylinder) | voxel grid: nx ny nz voxel_size
  float4 meta;        // .x = enabled (1.0 / 0.0; already includes o < count), .y = flat obstacle index (int bits),
                      // .z / .w = primitive radius / half length (capsule: half segment; cylinder: half height)
};
constexpr int kObsRecFloats = sizeof(ObsRec) / sizeof(float);

// Explicit fma chain: the same bits in every kernel and inlining context.  The swept cost is
// discontinuous where two consecutive sphere positions coincide in the obstacle frame (zero sweep
// length -> no sweep samples), so two code paths only agree on such points when this transform
// rounds identically in both (with -ffp-contract=fast the compiler picks the fusion per context).
__device__ __forceinline__ f3 to_local(const ObsRec &r, f3 v) {
  return make_f3(__builtin_fmaf(r.r0.x, v.x, __builtin_fmaf(r.r0.y, v.y, __builtin_fmaf(r.r0.z, v.z, r.r0.w))),
                 __builtin_fmaf(r.r1.x, v.x, __builtin_fmaf(r.r1.y, v.y, __builtin_fmaf(r.r1.z, v.z, r.r1.w))),
                 __builtin_fmaf(r.r2.x, v.x, __builtin_fmaf(r.r2.y, v.y, __builtin_fmaf(r.r2.z, v.z, r.r2.w))));
}
__device__ __forceinline__ f3 to_world_vector(const ObsRec &r, f3 v) {  // R^T v
  return make_f3(r.r0.x * v.x + r.r1.x * v.y + r.r2.x * v.z, r.r0.y * v.x + r.r1.y * v.y + r.r2.y * v.z,
                 r.r0.z * v.x + r.r1.z * v.y + r.r2.z * v.z);
}

// wp_collision_common.py:11-38
__device__ __forceinline__ void activation(float dist, float eta, float &cost, float &gscale) {
  if (dist > eta) { cost = dist - 0.5f * eta; gscale = 1.0f; }
  else { cost = 0.5f * dist * dist / eta; gscale = dist / eta; }
}

// data_cuboid.py:547-628; g = minus the SDF gradient
// (half = half extents; the gradient is only worked out by the caller's penetrating lanes)
__device__ __forceinline__ float cuboid_sdf(float4 half, f3 lp, bool want_grad, float r_adj, f3 &g) {
  const float qx = fabsf(lp.x) - half.x, qy = fabsf(lp.y) - half.y, qz = fabsf(lp.z) - half.z;
  const float cx = fmaxf(qx, 0.0f), cy = fmaxf(qy, 0.0f), cz = fmaxf(qz, 0.0f);
  const float od = sqrtf(cx * cx + cy * cy + cz * cz);
  const float mq = fmaxf(qx, fmaxf(qy, qz));
  const float sdf = od + fminf(mq, 0.0f);
  g = make_f3(0.f, 0.f, 0.f);
  if (!(want_grad && r_adj - sdf > 0.0f)) return sdf;
  if (od > 1e-6f) {
    const float inv = -1.0f / od;
    g = make_f3(cx * inv, cy * inv, cz * inv);
    if (lp.x < 0.0f) g.x = -g.x;
    if (lp.y < 0.0f) g.y = -g.y;
    if (lp.z < 0.0f) g.z = -g.z;
  } else {
    if (fabsf(qx - mq) < 1e-6f) g.x = (lp.x < 0.0f) ? 1.0f : -1.0f;
    else if (fabsf(qy - mq) < 1e-6f) g.y = (lp.y < 0.0f) ? 1.0f : -1.0f;
    else g.z = (lp.z < 0.0f) ? 1.0f : -1.0f;
  }
  return sdf;
}

// Analytic primitives in the cuboid store (tag in dims.w, see include/curobo_hip.h): the reference turns Sphere /
// Capsule / Cylinder obstacles into triangle meshes and queries them through Warp's BVH (geom/types.py:290-450,
// :1104-1124, geom/data/data_mesh.py:555-700); here they are their closed forms in the obstacle frame (axis = local
// z, centred at the origin: the trimesh.creation conventions the reference's get_trimesh_mesh uses).  g = minus the
// SDF gradient, as for cuboid_sdf; the same host-side fields are curobo_amd/scene/primitives.py.
__device__ __forceinline__ float primitive_sdf(const ObsRec &rec, f3 lp, bool want_grad, float r_adj, f3 &g) {
  const int tag = (int)rec.shape.w;
  const float r = rec.meta.z, hl = rec.meta.w;
  g = make_f3(0.f, 0.f, 0.f);
  if (tag == 3) {  // cylinder: radius r, half height hl
    const float rho = sqrtf(lp.x * lp.x + lp.y * lp.y);
    const float dr = rho - r, dz = fabsf(lp.z) - hl;
    const float cr = fmaxf(dr, 0.0f), cz = fmaxf(dz, 0.0f);
    const float od = sqrtf(cr * cr + cz * cz);
    const float sdf = od + fminf(fmaxf(dr, dz), 0.0f);
    if (!(want_grad && r_adj - sdf > 0.0f)) return sdf;
    const float ir = rho > 1e-6f ? 1.0f / rho : 0.0f;
    const f3 er = rho > 1e-6f ? make_f3(lp.x * ir, lp.y * ir, 0.0f) : make_f3(1.0f, 0.0f, 0.0f);
    const float sz = lp.z < 0.0f ? -1.0f : 1.0f;
    if (od > 1e-6f) g = make_f3(-er.x * cr / od, -er.y * cr / od, -sz * cz / od);
    else if (dr > dz) g = make_f3(-er.x, -er.y, 0.0f);
    else g = make_f3(0.0f, 0.0f, -sz);
    return sdf;
  }
  // sphere (hl = 0) and capsule: distance to the segment [-hl, hl] on the local z axis, minus r
  const float t = fminf(fmaxf(lp.z, -hl), hl);
  const f3 v = make_f3(lp.x, lp.y, lp.z - t);
  const float d = sqrtf(dot(v, v));
  const float sdf = d - r;
  if (!(want_grad && r_adj - sdf > 0.0f)) return sdf;
  if (d > 1e-6f) g = make_f3(-v.x / d, -v.y / d, -v.z / d);
  else g = make_f3(0.0f, 0.0f, -1.0f);
  return sdf;
}

// data_voxel.py:781-1056 + :1164-1215; g = normalised minus-gradient, 0 when sdf >= max_dist
__device__ __forceinline__ float voxel_sdf(const curobo_hip_scene &sc, int flat_idx, float4 prm, f3 lp, f3 &g) {
  const int nx = (int)prm.x, ny = (int)prm.y, nz = (int)prm.z;
  const float vs = prm.w, max_dist = sc.voxel_max_distance;
  const __half *feat = reinterpret_cast<const __half *>(sc.voxel_features) + (size_t)flat_idx * sc.voxel_n_voxels;
  float sdf, gx = 0.f, gy = 0.f, gz = 0.f;
  if (nx < 2 || ny < 2 || nz < 2) {
    const int ix = (int)((lp.x + (float)nx * vs * 0.5f) / vs);
    const int iy = (int)((lp.y + (float)ny * vs * 0.5f) / vs);
    const int iz = (int)((lp.z + (float)nz * vs * 0.5f) / vs);
    const bool ok = ix >= 0 && ix < nx && iy >= 0 && iy < ny && iz >= 0 && iz < nz;
    sdf = ok ? __half2float(feat[ix * ny * nz + iy * nz + iz]) : max_dist;
  } else {
    const float inv_voxel = 1.0f / vs;
    const float vx = lp.x * inv_voxel + (float)nx * 0.5f - 0.5f;
    const float vy = lp.y * inv_voxel + (float)ny * 0.5f - 0.5f;
    const float vz = lp.z * inv_voxel + (float)nz * 0.5f - 0.5f;
    const int x0 = (int)floorf(vx), y0 = (int)floorf(vy), z0 = (int)floorf(vz);
    const float fx = vx - (float)x0, fy = vy - (float)y0, fz = vz - (float)z0;
    const float fx1 = 1.0f - fx, fy1 = 1.0f - fy, fz1 = 1.0f - fz;
    const int sx = ny * nz, sy = nz;
    const bool x0ok = x0 >= 0 && x0 < nx, x1ok = x0 + 1 >= 0 && x0 + 1 < nx;
    const bool y0ok = y0 >= 0 && y0 < ny, y1ok = y0 + 1 >= 0 && y0 + 1 < ny;
    const bool z0ok = z0 >= 0 && z0 < nz, z1ok = z0 + 1 >= 0 && z0 + 1 < nz;
    const long base = (long)x0 * sx + (long)y0 * sy + z0;
    const bool ok[8] = {x0ok && y0ok && z0ok, x0ok && y0ok && z1ok, x0ok && y1ok && z0ok, x0ok && y1ok && z1ok,
                        x1ok && y0ok && z0ok, x1ok && y0ok && z1ok, x1ok && y1ok && z0ok, x1ok && y1ok && z1ok};
    const int off[8] = {0, 1, sy, sy + 1, sx, sx + 1, sx + sy, sx + sy + 1};
    float s[8];
    bool all_valid = true;
#pragma unroll
    for (int k = 0; k < 8; k++) all_valid = all_valid && ok[k];
    if (all_valid) {
      // interior: z is the fastest index, so the corners come as four (z0, z0 + 1) pairs = four 4-byte loads (2-byte
      // aligned: global memory takes unaligned dwords) instead of eight predicated 2-byte loads
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t w;
        __builtin_memcpy(&w, feat + base + off[2 * k], 4);
        s[2 * k] = __half2float(__ushort_as_half((unsigned short)(w & 0xffffu)));
        s[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = ok[k] ? __half2float(feat[base + off[k]]) : max_dist;
    }
    if (all_valid) {
      sdf = s[0] * fx1 * fy1 * fz1 + s[1] * fx1 * fy1 * fz + s[2] * fx1 * fy * fz1 + s[3] * fx1 * fy * fz +
            s[4] * fx * fy1 * fz1 + s[5] * fx * fy1 * fz + s[6] * fx * fy * fz1 + s[7] * fx * fy * fz;
      gx = ((s[4] - s[0]) * fy1 * fz1 + (s[5] - s[1]) * fy1 * fz + (s[6] - s[2]) * fy * fz1 + (s[7] - s[3]) * fy * fz) * inv_voxel;
      gy = ((s[2] - s[0]) * fx1 * fz1 + (s[3] - s[1]) * fx1 * fz + (s[6] - s[4]) * fx * fz1 + (s[7] - s[5]) * fx * fz) * inv_voxel;
      gz = ((s[1] - s[0]) * fx1 * fy1 + (s[3] - s[2]) * fx1 * fy + (s[5] - s[4]) * fx * fy1 + (s[7] - s[6]) * fx * fy) * inv_voxel;
    } else {
      const float wts[8] = {fx1 * fy1 * fz1, fx1 * fy1 * fz, fx1 * fy * fz1, fx1 * fy * fz,
                            fx * fy1 * fz1,  fx * fy1 * fz,  fx * fy * fz1,  fx * fy * fz};
      float wsum = 0.f, vsum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float v = ok[k] ? 1.0f : 0.0f;
        vsum += s[k] * wts[k] * v;
        wsum += wts[k] * v;
      }
      if (wsum <= 0.0f) {
        sdf = max_dist;
      } else {
        sdf = vsum / wsum;
        const float wx[4] = {fy1 * fz1, fy1 * fz, fy * fz1, fy * fz};
        const float wy[4] = {fx1 * fz1, fx1 * fz, fx * fz1, fx * fz};
        const float wz[4] = {fx1 * fy1, fx1 * fy, fx * fy1, fx * fy};
        float gs = 0.f, gw = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++)  // pairs (k, k+4)
          if (ok[k] && ok[k + 4]) { gs += (s[k + 4] - s[k]) * wx[k]; gw += wx[k]; }
        gx = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
        gs = gw = 0.f;
        const int py[4] = {0, 1, 4, 5};
#pragma unroll
        for (int k = 0; k < 4; k++)  // pairs (p, p+2)
          if (ok[py[k]] && ok[py[k] + 2]) { gs += (s[py[k] + 2] - s[py[k]]) * wy[k]; gw += wy[k]; }
        gy = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
        gs = gw = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++)  // pairs (2k, 2k+1)
          if (ok[2 * k] && ok[2 * k + 1]) { gs += (s[2 * k + 1] - s[2 * k]) * wz[k]; gw += wz[k]; }
        gz = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
      }
    }
  }
  g = make_f3(0.f, 0.f, 0.f);
  if (sdf >= max_dist) return max_dist;
  const f3 ng = make_f3(-gx, -gy, -gz);
  const float len = sqrtf(dot(ng, ng));
  if (len > 1e-6f) g = make_f3(ng.x / len, ng.y / len, ng.z / len);
  return sdf;
}

// penetration of one sample + its cost c and gradient scale gs * direction g (both 0 when it does not penetrate)
template <bool VOXEL, bool PRIMS = false>
__device__ __forceinline__ float point_terms(const curobo_hip_scene &sc, int flat, const ObsRec &rec, f3 lp, float r_adj,
                                             float eta, float &c, float &gs, f3 &g) {
  float sdf;
  if (VOXEL) sdf = voxel_sdf(sc, flat, rec.shape, lp, g);
  else if (PRIMS && rec.shape.w != 0.0f) sdf = primitive_sdf(rec, lp, true, r_adj, g);
  else sdf = cuboid_sdf(rec.shape, lp, true, r_adj, g);
  const float pen = -sdf + r_adj;
  c = 0.0f; gs = 0.0f;
  if (pen > 0.0f) activation(pen, eta, c, gs);
  return pen;
}

template <bool VOXEL, bool PRIMS = false>
__device__ __forceinline__ float eval_point(const curobo_hip_scene &sc, int flat, const ObsRec &rec, f3 lp, float r_adj,
                                            float eta, float &cost_sum, f3 &grad_sum) {
  f3 g;
  float c, gs;
  const float pen = point_terms<VOXEL, PRIMS>(sc, flat, rec, lp, r_adj, eta, c, gs, g);
  if (pen > 0.0f) {
    cost_sum += c;
    grad_sum = grad_sum + gs * g;
  }
  return pen;
}

// One obstacle record as the kernels consume it.  STAGED: the workgroup builds the records of the
// (few) batch rows it touches in LDS once, so the obstacle loop reads them as LDS broadcasts instead
// of a dependent global-memory round trip (and a quaternion expansion) per obstacle and sphere.
template <bool VOXEL>
__device__ __forceinline__ ObsRec load_rec_global(const curobo_hip_scene &sc, int env, int o) {
  const int max_n = VOXEL ? sc.max_voxel_grids : sc.max_cuboids;
  const int count = VOXEL ? sc.voxel_count[env] : sc.cuboid_count[env];
  const uint8_t *enable = VOXEL ? sc.voxel_enable : sc.cuboid_enable;
  const float *inv_pose = VOXEL ? sc.voxel_inv_pose : sc.cuboid_inv_pose;
  const float *shape = VOXEL ? sc.voxel_params : sc.cuboid_dims;
  const int flat = env * max_n + o;
  const float4 p = reinterpret_cast<const float4 *>(inv_pose)[(size_t)flat * 2];      // x y z qw
  const float4 q = reinterpret_cast<const float4 *>(inv_pose)[(size_t)flat * 2 + 1];  // qx qy qz pad
  const float x = q.x, y = q.y, z = q.z, w = p.w;
  // The rotation as EXPLICIT fused multiply-adds over single products.  Written with * and +, every kernel that inlines this
  // function contracts the nine expressions its own way (-ffp-contract=fast), and two kernels then hold rotations that differ
  // in the last bit: enough for a sphere that is stationary up to rounding to coincide with its neighbour in one kernel's
  // obstacle frame and not in the other's -- the sweep's duplicate centre sample (half_dist > 0) then exists in one of them
  // only (found by tests/randomised/fuzz_fused.py on randomly rotated cuboids: 19 of 1 496 such trajectories differed between
  // the fused launch and the kernel sequence on identical spheres).
  const float x2 = 2.0f * x, y2 = 2.0f * y, w2 = 2.0f * w;  // (exact)
  const float k = __builtin_fmaf(w2, w, -1.0f);
  const float wz = w2 * z, wy = w2 * y, wx = w2 * x;  // one rounding each, consumed by an fma below: nothing left to contract
  ObsRec r;
  r.r0 = make_float4(__builtin_fmaf(x2, x, k), __builtin_fmaf(x2, y, -wz), __builtin_fmaf(x2, z, wy), p.x);
  r.r1 = make_float4(__builtin_fmaf(x2, y, wz), __builtin_fmaf(y2, y, k), __builtin_fmaf(y2, z, -wx), p.y);
  r.r2 = make_float4(__builtin_fmaf(x2, z, -wy), __builtin_fmaf(y2, z, wx), __builtin_fmaf(2.0f * z, z, k), p.z);
  r.shape = reinterpret_cast<const float4 *>(shape)[flat];
  float prim_r = 0.0f, prim_hl = 0.0f;
  if (!VOXEL) {
    const float4 d = r.shape;
    const int tag = (int)d.w;  // 0 = cuboid (the reference's dims[..., 3] padding is zero)
    if (tag == 0) {
      r.shape = make_float4(d.x * 0.5f, d.y * 0.5f, d.z * 0.5f, 0.0f);
    } else {  // primitive: (radius, half length, -, tag); .xyz becomes its bounding box for the early rejects
      prim_r = d.x;
      prim_hl = tag == 1 ? 0.0f : d.y;
      r.shape = make_float4(d.x, d.x, tag == 1 ? d.x : (tag == 2 ? d.y + d.x : d.y), (float)tag);
    }
  }
  // .x = is_obs_enabled (data_cuboid.py:467-485), .y = flat obstacle index env * max_n + o (integer bits)
  r.meta = make_float4((o < count && enable[flat] == 1) ? 1.0f : 0.0f, __int_as_float(flat), prim_r, prim_hl);
  return r;
}

// Cost and obstacle-frame gradient of ONE sphere against ONE obstacle that passed the early reject:
// the centre sample plus the sweep towards the previous / next point (the body of obstacle_set's
// loop; also the unit of work of the fused kernel's scene pass).  lc = centre in the obstacle frame.
template <bool VOXEL, int SWEEP, bool PRIMS = false>
__device__ __forceinline__ void obstacle_contribution(const curobo_hip_scene &sc, const ObsRec &rec, int flat, f3 lc,
                                                      bool has_prev, bool has_next, f3 prev_c, f3 next_c, float r_adj, float eta,
                                                      float half_w_prev, float half_w_next, float &cost_sum, f3 &grad_local) {
  f3 g_c;
  float c_c, gs_c;
  const float pen_c = point_terms<VOXEL, PRIMS>(sc, flat, rec, lc, r_adj, eta, c_c, gs_c, g_c);
  if (pen_c > 0.0f) {
    cost_sum += c_c;
    grad_local = grad_local + gs_c * g_c;
  }
  if (SWEEP > 0) {  // wp_sweep_collision_kernel.py:176-254
    // outside a voxel grid the SDF is the constant max_dist: no bound across the grid face
    const float sdf_c = r_adj - pen_c;
    const bool can_cull = VOXEL ? (sdf_c < sc.voxel_max_distance) : true;
    // voxel slack: interpolated values are convex combinations of corner samples that sit within
    // sqrt(3) voxels of the query, once at the centre and once at the sample, + fp16 rounding
    const float slack = VOXEL ? 3.5f * rec.shape.w + 0.002f * fabsf(sdf_c) : 0.0f;
    const float clearance = -pen_c;
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      const float half_w = dir == 0 ? half_w_prev : half_w_next;
      const bool culled = can_cull && clearance > half_w * 1.0001f + slack + 1e-6f;
      if ((dir == 0 ? has_prev : has_next) && !culled) {
        const f3 ln = to_local(rec, dir == 0 ? prev_c : next_c);
        const f3 dd = ln - lc;
        const float half_dist = sqrtf(dot(dd, dd)) * 0.5f;
        const float inv_half = 1.0f / fmaxf(half_dist, 0.001f);
        float jump = 0.0f;
        // The reference's first sample of a direction (k = 0) sits at jump = 0, i.e. t = 1: the centre itself
        // (lp = 1 * lc + 0 * ln).  Its signed distance, cost and gradient are the centre sample's, bit for bit, so they
        // are added again instead of being evaluated again (for an ESDF: eight gathers fewer, per direction, at the
        // head of the serial chain of the sweep).
        if (!(jump >= half_dist)) {
          if (pen_c > 0.0f) {
            cost_sum += c_c;
            grad_local = grad_local + gs_c * g_c;
            jump += pen_c;
          } else if (-pen_c >= 1000.0f) jump += r_adj;
          else jump += fmaxf(-pen_c, r_adj);
          for (int k = 1; k < SWEEP; k++) {
            if (jump >= half_dist) break;
            const float tt = 1.0f - 0.5f * jump * inv_half;
            const f3 lp = tt * lc + (1.0f - tt) * ln;
            const float p2 = eval_point<VOXEL, PRIMS>(sc, flat, rec, lp, r_adj, eta, cost_sum, grad_local);
            if (p2 > 0.0f) jump += p2;
            else if (-p2 >= 1000.0f) jump += r_adj;
            else jump += fmaxf(-p2, r_adj);
          }
        }
      }
    }
  }
}

// Early reject of one obstacle for one sphere (see obstacle_set): true = contributes exactly zero.
// reach = max half sweep length * 1.0001 + 2e-6 (2e-6 without sweep), thr2_c = ((r_adj + reach)^2) * 1.00001.
template <bool VOXEL>
__device__ __forceinline__ bool obstacle_early_reject(const curobo_hip_scene &sc, const ObsRec &rec, f3 lc, float r_adj,
                                                      float reach, float thr2_c) {
  if (!VOXEL) {
    const float cx = fmaxf(fabsf(lc.x) - rec.shape.x, 0.0f), cy = fmaxf(fabsf(lc.y) - rec.shape.y, 0.0f),
                cz = fmaxf(fabsf(lc.z) - rec.shape.z, 0.0f);
    return cx * cx + cy * cy + cz * cz > thr2_c;
  } else if (r_adj < sc.voxel_max_distance) {
    const float vs = rec.shape.w;
    const float cx = fmaxf(fabsf(lc.x) - rec.shape.x * vs * 0.5f, 0.0f), cy = fmaxf(fabsf(lc.y) - rec.shape.y * vs * 0.5f, 0.0f),
                cz = fmaxf(fabsf(lc.z) - rec.shape.z * vs * 0.5f, 0.0f);
    const float thr_v = reach + vs;
    if (cx * cx + cy * cy + cz * cz > thr_v * thr_v * 1.00001f) return true;
    // Coarse minimum (scene upload, optional): every sample of this sphere -- centre and sweep -- interpolates
    // corner voxels within ceil(reach / vs) + 1 voxels of the centre's voxel, a convex combination of values
    // >= the dilated block minimum m (out-of-grid corners read max_distance >= m as well).  m > r_adj
    // therefore means penetration = r_adj - sdf < 0 at every sample: exactly zero cost and gradient.
    if (sc.voxel_coarse_min != nullptr && reach <= (float)(sc.voxel_coarse_dilate - 1) * vs) {
      const int nx = (int)rec.shape.x, ny = (int)rec.shape.y, nz = (int)rec.shape.z, blk = sc.voxel_coarse_block;
      const float inv = 1.0f / vs;
      const int ix = (int)floorf(lc.x * inv + (float)nx * 0.5f), iy = (int)floorf(lc.y * inv + (float)ny * 0.5f),
                iz = (int)floorf(lc.z * inv + (float)nz * 0.5f);
      if (ix >= 0 && ix < nx && iy >= 0 && iy < ny && iz >= 0 && iz < nz) {
        const int cy = (ny + blk - 1) / blk, cz = (nz + blk - 1) / blk;
        const size_t at = (size_t)__float_as_int(rec.meta.y) * sc.voxel_n_coarse + ((size_t)(ix / blk) * cy + iy / blk) * cz + iz / blk;
        const float m = __half2float(reinterpret_cast<const __half *>(sc.voxel_coarse_min)[at]);
        if (m * 0.9999f > r_adj + 1e-6f) return true;
      }
    }
  }
  return false;
}

// Sweep culling (result-preserving): every swept sample lies within half_dist of the current
// centre (t in (0.5, 1]) and a signed distance field is 1-Lipschitz, so when the centre's
// clearance  sdf - r_adj  exceeds half_dist (+ an interpolation slack for voxel grids) no sample
// can penetrate and the whole sweep direction contributes exactly zero.
// half_w* are the world-frame half segment lengths (rigid transforms preserve them).
// Early reject (result-preserving as well): thr2 = (r_adj + max half sweep length + margin)^2.  A
// cuboid whose squared outside distance exceeds it is clear of the centre by more than any sweep
// can reach: no penetration, both sweep directions culled -> exactly zero, without sqrt, gradient
// or sweep bookkeeping.  A voxel grid is skipped when the centre is so far outside its box (sweep
// reach + one voxel) that every sample reads the constant max_distance.
template <bool VOXEL, int SWEEP, bool STAGED, bool PRIMS = false>
__device__ __forceinline__ void obstacle_set(const curobo_hip_scene &sc, const ObsRec *__restrict__ recs, int env,
                                             bool has_prev, bool has_next, f3 prev_c, f3 next_c, f3 center, float r_adj,
                                             float eta, float w, float half_w_prev, float half_w_next, uint32_t mask,
                                             float &dsum, f3 &gsum) {
  const int max_n = VOXEL ? sc.max_voxel_grids : sc.max_cuboids;
  const int bit0 = VOXEL ? sc.max_cuboids : 0;
  const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
  const float thr_c = r_adj + reach;
  const float thr2_c = thr_c * thr_c * 1.00001f;
  for (int o = 0; o < max_n; o++) {
    if (bit0 + o < 32 && !((mask >> (bit0 + o)) & 1u)) continue;  // culled by the caller's bounding volume
    const ObsRec rec = STAGED ? recs[o] : load_rec_global<VOXEL>(sc, env, o);
    if (rec.meta.x == 0.0f) continue;
    const int flat = env * max_n + o;
    const f3 lc = to_local(rec, center);
    if (obstacle_early_reject<VOXEL>(sc, rec, lc, r_adj, reach, thr2_c)) continue;
    float cost_sum = 0.0f;
    f3 grad_local = make_f3(0.f, 0.f, 0.f);
    obstacle_contribution<VOXEL, SWEEP, PRIMS>(sc, rec, flat, lc, has_prev, has_next, prev_c, next_c, r_adj, eta, half_w_prev,
                                               half_w_next, cost_sum, grad_local);
    if (cost_sum > 0.0f) {
      const f3 gw = to_world_vector(rec, grad_local);
      dsum += w * cost_sum;
      gsum = gsum + w * gw;
    }
  }
}

// Which of the first 32 obstacles can touch ANY sphere inside a bounding ball (world centre C,
// radius R >= |c_s - C| + r_s for every contained sphere) whose spheres sweep at most `reach`
// (>= the half sweep length of each of them)?  A signed distance field is 1-Lipschitz, so
// sdf(c_s) - r_s >= sdf(C) - R; when that exceeds eta + reach the per-sphere test above would
// reject the obstacle for every contained sphere, i.e. clearing its bit is result-preserving.
// Voxel grids: cleared when the ball (+ reach + one voxel) lies outside the grid box, where every
// sample reads the constant max_distance.  Obstacles >= 32 are never culled.
template <int KINDS>
__device__ __forceinline__ uint32_t bounding_ball_obstacle_mask(const curobo_hip_scene &sc, const ObsRec *__restrict__ recs,
                                                                f3 C, float R, float eta, float reach) {
  uint32_t mask = 0u;
  const float thr = (R + eta + reach) * 1.0001f + 4e-6f;
  const float thr2 = thr * thr * 1.00001f;
  const int n_c = (KINDS & 1) ? sc.max_cuboids : 0, n_v = (KINDS & 2) ? sc.max_voxel_grids : 0;
  for (int o = 0; o < n_c + n_v && o < 32; o++) {
    const ObsRec rec = recs[o < n_c ? o : sc.max_cuboids + (o - n_c)];
    const int bit = o < n_c ? o : sc.max_cuboids + (o - n_c);
    if (bit >= 32) break;
    if (rec.meta.x == 0.0f) continue;
    const f3 lc = to_local(rec, C);
    float hx = rec.shape.x, hy = rec.shape.y, hz = rec.shape.z, t2 = thr2;
    if (o >= n_c) {  // voxel grid: box half extents, one voxel of margin, only when no radius reaches max_distance
      const float vs = rec.shape.w;
      hx *= vs * 0.5f; hy *= vs * 0.5f; hz *= vs * 0.5f;
      const float tv = (R + reach + vs) * 1.0001f + 4e-6f;
      t2 = (R + eta < sc.voxel_max_distance) ? tv * tv * 1.00001f : 3.0e38f;
    }
    const float cx = fmaxf(fabsf(lc.x) - hx, 0.0f), cy = fmaxf(fabsf(lc.y) - hy, 0.0f), cz = fmaxf(fabsf(lc.z) - hz, 0.0f);
    if (!(cx * cx + cy * cy + cz * cz > t2)) mask |= 1u << bit;
  }
  return mask;
}

// Speed metric (wp_speed_metric.py:38-93) of a sphere with both neighbours: cost and gradient are
// scaled by the sphere's speed and the gradient is projected off the velocity direction (minus the
// curvature term).  The map is linear in (dsum, gsum), so it may be applied per obstacle contribution.
__device__ __forceinline__ void speed_metric_apply(f3 center, f3 pp, f3 np, float speed_dt, float &dsum, f3 &gsum) {
  float dt = speed_dt;
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

// Full scene cost of ONE sphere of one trajectory point: every enabled obstacle (cuboids, then voxel
// grids, in index order), optional sweep towards the previous / next point and the fused speed
// metric (wp_speed_metric.py:38-93).  KINDS: bit 0 = cuboid store present, bit 1 = voxel grids, bit 2 = the cuboid
// store may hold analytic primitives (sphere / capsule / cylinder tags).
template <int SWEEP, bool STAGED, int KINDS>
__device__ __forceinline__ void sphere_scene_cost(const curobo_hip_scene &sc, const ObsRec *__restrict__ recs, int env,
                                                  float4 s, bool has_prev, float4 ps, bool has_next, float4 ns, float eta,
                                                  float w, bool speed_metric, float speed_dt, float &dsum, f3 &gsum,
                                                  uint32_t mask = 0xffffffffu) {
  dsum = 0.0f;
  gsum = make_f3(0.f, 0.f, 0.f);
  const f3 center = make_f3(s.x, s.y, s.z);
  const f3 pp = make_f3(ps.x, ps.y, ps.z), np = make_f3(ns.x, ns.y, ns.z);
  if (s.w >= 0.0f) {
    const float r_adj = s.w + eta;
    float half_w_prev = 0.0f, half_w_next = 0.0f;
    if (SWEEP > 0) {
      if (has_prev) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (has_next) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    if (KINDS & 1)
      obstacle_set<false, SWEEP, STAGED, (KINDS & 4) != 0>(sc, recs, env, has_prev, has_next, pp, np, center, r_adj, eta, w,
                                                           half_w_prev, half_w_next, mask, dsum, gsum);
    if (KINDS & 2)
      obstacle_set<true, SWEEP, STAGED>(sc, recs + sc.max_cuboids, env, has_prev, has_next, pp, np, center, r_adj, eta, w,
                                        half_w_prev, half_w_next, mask, dsum, gsum);
  }
  if (speed_metric && has_prev && has_next && dsum > 0.0f) speed_metric_apply(center, pp, np, speed_dt, dsum, gsum);
}

}  // namespace curobo_hip
