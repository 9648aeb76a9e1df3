// mesh_device.hpp -- device functions of the triangle-mesh obstacle kind: the closest-point / sign query through the
// linear BVH of mesh_bvh.hip and the cost of one sphere against one mesh (centre + sweep), shared by the mesh launch
// (mesh_bvh.hip) and the fused rollout kernels (rollout_fused.hip).
//
// Reference: curobo/_src/geom/data/data_mesh.py:555-700 (wp.mesh_query_point per query sphere: closest surface point,
// signed distance, unit vector to the point) and geom/collision/wp_sweep_collision_kernel.py:176-254 (the sweep).  What
// is restated is the published contract of mesh_query_point, not Warp's BVH.
//
// Traversal.  The tree is a complete binary tree in heap layout (node k -> 2k, 2k + 1; mesh_bvh.hip), so a walk needs
// no stack: the path is the node index itself.  A lane keeps two bit trails indexed by depth -- which child of a node
// it entered first (the nearer one) and whether the other child is still owed a visit -- and finds the next node by
// shifting the index.  Everything lives in a handful of VGPRs: the private `int stack[64]` of the first version was
// 256 B of scratch per lane and the reason the launch ran at 0.005 of the HBM roofline.
#pragma once
#include "common.hpp"
#include "self_device.hpp"

namespace curobo_hip {

#ifdef CUROBO_MESH_STATS  // diagnostic builds only (tools/r04/mesh_stats.py): work counters of the walks
__device__ unsigned long long g_mesh_stats[8];  // closest calls, nodes tested, triangles, ray walks, ray nodes, full queries, items
__device__ unsigned int g_mesh_lane[1 << 18];   // per queue entry: moves [0, 2^17), transitions [2^17, 2^18)
#define CUROBO_MESH_COUNT(i, n)                                                                          \
  do {                                                                                                   \
    atomicAdd(&g_mesh_stats[i], (unsigned long long)(n));                                                \
  } while (0)
#else
#define CUROBO_MESH_COUNT(i, n) do {} while (0)
#endif

struct TriRec {  // 48 bytes
  float4 a, ab, ac;
};

__device__ __forceinline__ float box_dist2(const float4 lo, const float4 hi, f3 p) {
  const float dx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.0f), dy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.0f),
              dz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.0f);
  return dx * dx + dy * dy + dz * dz;
}

// closest point of triangle (a, a + ab, a + ac) to p (Ericson, Real-Time Collision Detection 5.1.5)
// region: 0 = the face, 1 / 2 / 3 = vertex a / b / c, 4 / 5 / 6 = edge ab / bc / ca (the order of curobo_hip_mesh.tri_pn, + 1)
// Without branches: the lanes of a wavefront meet different regions, and a chain of early returns is executed as the union of
// its paths with the exec-mask bookkeeping of every test.  Every region's test is evaluated, the first that holds in
// Ericson's order (a, b, ab, c, ca, bc, face) is taken, the three edge cases share one division.
__device__ __forceinline__ f3 closest_on_triangle(f3 p, f3 a, f3 ab, f3 ac, int &region) {
  const f3 ap = p - a;
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  const f3 b = a + ab, bp = p - b;
  const float d3 = dot(ab, bp), d4 = dot(ac, bp);
  const f3 c = a + ac, cp = p - c;
  const float d5 = dot(ab, cp), d6 = dot(ac, cp);
  const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  const float e43 = d4 - d3, e56 = d5 - d6;
  const bool at_a = d1 <= 0.0f && d2 <= 0.0f;
  const bool at_b = d3 >= 0.0f && d4 <= d3;
  const bool on_ab = vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f;
  const bool at_c = d6 >= 0.0f && d5 <= d6;
  const bool on_ca = vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f;
  const bool on_bc = va <= 0.0f && e43 >= 0.0f && e56 >= 0.0f;
  region = at_a ? 1 : at_b ? 2 : on_ab ? 4 : at_c ? 3 : on_ca ? 6 : on_bc ? 5 : 0;
  // the edge point: base + (num / den) dir
  const float num = on_ab ? d1 : on_ca ? d2 : e43;
  const float den = on_ab ? d1 - d3 : on_ca ? d2 - d6 : e43 + e56;
  const bool from_a = on_ab || on_ca;
  const f3 dir = on_ab ? ab : on_ca ? ac : c - b;
  const f3 edge = (from_a ? a : b) + (num / den) * dir;
  const float inv = 1.0f / (va + vb + vc);
  const f3 face = a + (vb * inv) * ab + (vc * inv) * ac;  // the projection on the triangle's plane
  const f3 vert = at_a ? a : at_b ? b : c;
  const bool is_vert = region >= 1 && region <= 3, is_face = region == 0;
  return is_vert ? vert : is_face ? face : edge;
}

// ---- the closest-point walk: nearer child first, no stack.  The tree is complete and in heap layout, so the node index IS
// the path; two bit trails indexed by depth record which child of a node was entered first and whether the other one is
// still owed a visit.  (A walk over four grandchildren per step -- half the depth, four independent box loads -- tested
// fewer nodes, 5.0 M instead of 7.4 M on the C2 mesh world, and was slower: the step's selection logic costs more than
// the loads it saves.)
__device__ __forceinline__ int mesh_walk_next(int node, uint32_t first, uint32_t &owed) {
  while (node > 1) {
    const int parent = node >> 1;
    const uint32_t bit = 1u << (30 - __builtin_clz(node));  // depth of the parent
    const int entered_first = (parent << 1) | ((first & bit) ? 1 : 0);
    if (node == entered_first && (owed & bit)) {
      owed &= ~bit;
      return node ^ 1;
    }
    node = parent;
  }
  return 0;
}

// which side of the surface p is on, judged by the feature its closest point lies on (see mesh_closest_point)
__device__ __forceinline__ int mesh_feature_side(const curobo_hip_mesh &m, f3 p, f3 cp, float best_d2, int best_t, int best_region, bool tie) {
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  int side = 0;
  const f3 d = p - cp;
  const TriRec r = tri[best_t];
  if (best_region == 0 && !tie) {
    // the face the closest point lies in: only trusted when the point is clearly off its plane
    const float sd = dot(d, cross(make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z)));
    side = (best_d2 > 1e-12f && sd != 0.0f) ? (sd > 0.0f ? 1 : -1) : 0;
  } else if (m.tri_pn != nullptr && best_region != 0 && best_d2 > 1e-12f) {
    // an edge or a vertex: its pseudonormal (for a closed, consistently oriented surface the sign of d . n is the sign of
    // the distance whatever the dihedral angles); a degenerate pseudonormal gives no verdict
    const float4 n4 = reinterpret_cast<const float4 *>(m.tri_pn)[(size_t)best_t * 6 + (best_region - 1)];
    const f3 n = make_f3(n4.x, n4.y, n4.z);
    const float sd = dot(d, n), scale = sqrtf(dot(n, n) * best_d2);
    side = fabsf(sd) > 1e-4f * scale ? (sd > 0.0f ? 1 : -1) : 0;
  }
  return side;
}

// closest surface point within sqrt(best_d2) of p; returns false when there is none
// side: +1 / -1 = p is on the outer / inner side of the surface as the feature the closest point lies on says (the face's
// normal, or the pseudonormal of the edge / vertex when the mesh carries them), 0 = no verdict (count crossings)
//
// "while-while" form: every lane first walks inner nodes until it holds a leaf (or is done), THEN the leaves are tested --
// the wavefront runs the long triangle code once per round with most lanes active, instead of in every iteration in
// which any of its 64 lanes happened to reach a leaf.
__device__ __forceinline__ bool mesh_closest_point(const curobo_hip_mesh &m, f3 p, float &best_d2, f3 &cp, int &side) {
  side = 0;
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  bool found = false, tie = false;
  int best_t = 0, best_region = 0;
  uint32_t first = 0u, owed = 0u;
  int node = 1;
  CUROBO_MESH_COUNT(0, 1);
  bool check_own = true;  // the node was reached as the root or as a sibling: its own box is still to be tested
  for (;;) {
    int leaf = -1;
    while (node != 0 && leaf < 0) {
      CUROBO_MESH_COUNT(1, 1);
      if (check_own && box_dist2(box[node * 2], box[node * 2 + 1], p) > best_d2) {
        node = mesh_walk_next(node, first, owed);
        continue;
      }
      if (node >= m.n_leaves) {
        leaf = node;
        node = mesh_walk_next(node, first, owed);  // (its own box is tested when it is reached: the leaf may shrink the distance)
        check_own = true;
        continue;
      }
      // both children's boxes are one 64-byte line: test them here, enter the nearer one, owe the other a visit
      const int c0 = node * 2;
      const float4 lo0 = box[c0 * 2], hi0 = box[c0 * 2 + 1], lo1 = box[c0 * 2 + 2], hi1 = box[c0 * 2 + 3];
      const float d0 = box_dist2(lo0, hi0, p), d1 = box_dist2(lo1, hi1, p);
      const bool in0 = d0 <= best_d2, in1 = d1 <= best_d2;
      if (!(in0 || in1)) {
        node = mesh_walk_next(node, first, owed);
        check_own = true;
        continue;
      }
      const uint32_t bit = 1u << (31 - __builtin_clz(node));  // depth of this node
      // nearer box first; a point INSIDE both boxes (every query from inside the surface, near the root) has distance
      // zero to both: then the box whose centre is nearer (a blind choice finds a far triangle first and prunes nothing)
      bool right_first = in1 && (!in0 || d1 < d0);
      if (in0 && in1 && d0 == d1) {
        const f3 c0v = make_f3(lo0.x + hi0.x, lo0.y + hi0.y, lo0.z + hi0.z) - 2.0f * p;
        const f3 c1v = make_f3(lo1.x + hi1.x, lo1.y + hi1.y, lo1.z + hi1.z) - 2.0f * p;
        right_first = dot(c1v, c1v) < dot(c0v, c0v);
      }
      first = right_first ? (first | bit) : (first & ~bit);
      owed = (in0 && in1) ? (owed | bit) : (owed & ~bit);
      node = c0 + (right_first ? 1 : 0);
      check_own = false;  // just tested
    }
    if (leaf < 0) break;
    const int t0 = (leaf - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
    CUROBO_MESH_COUNT(2, t1 - t0);
    for (int t = t0; t < t1; t++) {
      const TriRec r = tri[t];
      int region;
      const f3 c = closest_on_triangle(p, make_f3(r.a.x, r.a.y, r.a.z), make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z), region);
      const f3 d = p - c;
      const float d2 = dot(d, d);
      if (d2 <= best_d2) {
        // (the same distance from two triangles: the closest point lies on a feature they share -- or on two separate
        // ones; a face verdict is then not trusted)
        tie = found && d2 == best_d2;
        best_d2 = d2; cp = c; found = true; best_t = t; best_region = region;
      }
    }
  }
  if (found) side = mesh_feature_side(m, p, cp, best_d2, best_t, best_region, tie);
  return found;
}

// ---- the same walk by EIGHT LANES per query (mesh_contribution_group below: an aligned group of a wavefront; all eight hold
// the same query and leave with the same result).  A step covers three tree levels: the eight great-grandchildren of a node are adjacent in the heap
// layout (one 256-byte run of boxes), lane j tests descendant j, the group's in-range mask is a slice of the wavefront
// ballot, the nearest one is entered first and the others are owed a visit (eight bits per step level in a 64-bit trail);
// at a leaf lane j tests triangles j, j + 8, ...  One query per lane is bound by the instruction stream of 64 walkers of
// very different lengths sharing a wavefront (107 k instructions per wavefront for queries that need ~70 node visits,
// sphere_mesh_walk_kernel); eight lanes per query give eight times the wavefronts for the same items and walks a third as
// deep.  A leaf node here is any node >= n_leaves; the tree depth need not be a multiple of three (the last step is
// narrower).
template <int G>
__device__ __forceinline__ float group_min(float v) {  // G = 4, 8, 16 or 32 aligned lanes
  v = fminf(v, dpp_f<0xB1>(v));
  v = fminf(v, dpp_f<0x4E>(v));
  if (G >= 8) v = fminf(v, dpp_f<0x141>(v));  // row_half_mirror: lane i <-> 7 - i of its group of eight
  if (G >= 16) v = fminf(v, dpp_f<0x140>(v));  // row_mirror: i <-> 15 - i
  if (G >= 32) v = fminf(v, __shfl_xor(v, 16, 64));
  return v;
}
template <int G>
__device__ __forceinline__ unsigned group_ballot(bool pred) {
  return (unsigned)(__ballot(pred) >> (threadIdx.x & (64u - G))) & (G >= 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u));
}
// ``keys`` (mesh_contribution_group): the calling lane's column of a [MESH_GROUP_LEVELS][blockDim.x] float table in LDS (the
// ordering key of this lane's descendant at every step level: siblings are revisited nearest first and re-judged against the
// distance found since, without fetching their boxes again).
#define MESH_GROUP_LEVELS 8
#ifndef MESH_PREV_BOUND
#define MESH_PREV_BOUND 1
#endif

// crossings of the ray p + t d (t > 0) with the surface (fixed-order stackless walk: the order does not matter here)
__device__ __forceinline__ int mesh_ray_crossings(const curobo_hip_mesh &m, f3 p, f3 d) {
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  const f3 inv = make_f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int hits = 0;
  unsigned node = 1u;
  CUROBO_MESH_COUNT(3, 1);
  while (node != 0u) {
    CUROBO_MESH_COUNT(4, 1);
    const float4 lo = box[node * 2], hi = box[node * 2 + 1];
    // slab test (an empty padding box has lo > hi: t_enter > t_exit)
    const float tx0 = (lo.x - p.x) * inv.x, tx1 = (hi.x - p.x) * inv.x, ty0 = (lo.y - p.y) * inv.y, ty1 = (hi.y - p.y) * inv.y,
                tz0 = (lo.z - p.z) * inv.z, tz1 = (hi.z - p.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
    const float t_out = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const bool hit = (t_in <= t_out) && !(lo.x > hi.x);
    if (hit && node < (unsigned)m.n_leaves) { node = node * 2u; continue; }
    if (hit) {
      const int t0 = ((int)node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
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
    }
    // next node in pre-order: leave every subtree this node is the right end of, then step to the sibling
    node >>= __builtin_ctz(~node);
    node = node ? (node | 1u) : 0u;
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

// The reference's inside test as published (Warp's mesh_query_point -> mesh_query_inside, warp/native/mesh.h; the call is
// data_mesh.py:632,682): three rays from p along +x, +y, +z; a ray's NEAREST hit tells whether it met the front or the back
// of a face; inside iff all three rays hit and all three nearest hits are back faces.  On a closed, consistently oriented
// surface that is the function mesh_feature_side evaluates without a ray; meshes that are not (curobo_hip_mesh.sign_rule =
// 1) are signed with this walk: pre-order, pruned by the axis slab and the nearest hit so far.
__device__ __forceinline__ bool mesh_inside_warp_rays(const curobo_hip_mesh &m, f3 p) {
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  int votes = 0;
#pragma unroll 1
  for (int axis = 0; axis < 3; axis++) {
    const f3 d = make_f3(axis == 0 ? 1.0f : 0.0f, axis == 1 ? 1.0f : 0.0f, axis == 2 ? 1.0f : 0.0f);
    float best_t = 3.0e38f;
    bool back = false;
    unsigned node = 1u;
    while (node != 0u) {
      const float4 lo = box[node * 2], hi = box[node * 2 + 1];
      // the ray runs along one axis: inside the box's extent on the other two, and the box not behind the origin nor
      // beyond the nearest hit (an empty padding box has lo > hi and fails the first test)
      const bool in_x = p.x >= lo.x && p.x <= hi.x, in_y = p.y >= lo.y && p.y <= hi.y, in_z = p.z >= lo.z && p.z <= hi.z;
      const float pa = axis == 0 ? p.x : axis == 1 ? p.y : p.z, la = axis == 0 ? lo.x : axis == 1 ? lo.y : lo.z,
                  ha = axis == 0 ? hi.x : axis == 1 ? hi.y : hi.z;
      const bool lateral = axis == 0 ? (in_y && in_z) : axis == 1 ? (in_x && in_z) : (in_x && in_y);
      const bool hit = lateral && ha >= pa && (la - pa) <= best_t;
      if (hit && node < (unsigned)m.n_leaves) { node = node * 2u; continue; }
      if (hit) {
        const int t0 = ((int)node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
        for (int t = t0; t < t1; t++) {  // Moeller-Trumbore
          const TriRec r = tri[t];
          const f3 ab = make_f3(r.ab.x, r.ab.y, r.ab.z), ac = make_f3(r.ac.x, r.ac.y, r.ac.z);
          const f3 pv = cross(d, ac);
          const float det = dot(ab, pv);
          if (det == 0.0f) continue;
          const float idet = 1.0f / det;
          const f3 tv = p - make_f3(r.a.x, r.a.y, r.a.z);
          const float u = dot(tv, pv) * idet;
          if (u < 0.0f || u > 1.0f) continue;
          const f3 qv = cross(tv, ab);
          const float v = dot(d, qv) * idet;
          if (v < 0.0f || u + v > 1.0f) continue;
          const float tt = dot(ac, qv) * idet;
          // det = ab . (d x ac) = -d . (ab x ac): negative when the ray leaves through the back of the face
          if (tt > 0.0f && tt < best_t) { best_t = tt; back = det < 0.0f; }
        }
      }
      node >>= __builtin_ctz(~node);
      node = node ? (node | 1u) : 0u;
    }
    if (best_t < 3.0e38f && back) votes++;
  }
  return votes == 3;
}

// inside / outside of a query whose closest feature gave the verdict `side` (0 = none), by the mesh's rule
__device__ __forceinline__ bool mesh_point_inside(const curobo_hip_mesh &m, f3 p, int side) {
  if (m.sign_rule == 1) return mesh_inside_warp_rays(m, p);
  return side != 0 ? side < 0 : mesh_inside(m, p);
}

// The same signed distance, searched only as far as the caller can use it.  The cost of a sample needs the distance when
// it is below r_adj; the sweep's step needs it when it is below r_adj + what is left of the half segment -- beyond that
// the sample contributes nothing and the walk along the segment ends, whatever the exact value (result preserving).  So
// the closest point is looked for within `radius` only: the traversal then prunes almost every node for a sphere that is
// merely near the mesh's bounding box.  Nothing within `radius` means "outside, farther than radius" (returns
// max_distance) unless the point may lie INSIDE the surface deeper than `radius`: may_be_inside (the caller knows the
// point is not -- e.g. a sweep sample within half_dist of a centre that is outside) and the root box gate a second walk
// without a radius, whose closest feature then says which side the point is on.  ONE walk site (a retry loop): inlined
// copies of the walk are what the register count of the callers is made of.
__device__ __forceinline__ float mesh_sdf_within(const curobo_hip_mesh &m, f3 lp, float radius, float max_distance, bool may_be_inside,
                                                 f3 &g) {
  g = make_f3(0.f, 0.f, 0.f);
  bool full = !(radius < max_distance);
  float d2 = 0.0f;
  f3 cp = lp;
  int side = 0;
  bool found = false;
#pragma unroll 1
  for (int attempt = 0; attempt < 2; attempt++) {
    const float r = full ? max_distance : radius;
    d2 = r * r;
    found = mesh_closest_point(m, lp, d2, cp, side);
    if (found || full || !may_be_inside) break;
    const float *rb = m.node_box + 8;  // root box: a point outside it is outside the (closed) surface
    if (lp.x < rb[0] || lp.y < rb[1] || lp.z < rb[2] || lp.x > rb[4] || lp.y > rb[5] || lp.z > rb[6]) break;
    full = true;  // inside the box, nothing within the radius: far outside in a concavity -- or deep inside
  }
  if (!found) return max_distance;
  const float d = sqrtf(d2);
  const f3 delta = lp - cp;
  if (d > 1e-6f) g = (1.0f / d) * delta;
  // the feature the closest point lies on says which side the point is on; without a verdict the crossings are counted
  const bool inside = mesh_point_inside(m, lp, side);
  return inside ? -d : d;
}

// data_mesh.py:630-700 compute_local_sdf_with_grad: signed distance (negative inside) and the local gradient
// (p - closest) / |p - closest| -- as the reference returns it, whatever side p is on.  No surface within max_distance:
// (max_distance, 0).
__device__ __forceinline__ float mesh_sdf_with_grad(const curobo_hip_mesh &m, f3 lp, float max_distance, f3 &g) {
  return mesh_sdf_within(m, lp, max_distance, max_distance, false, g);
}

__device__ __forceinline__ void activation_m(float dist, float eta, float &cost, float &gscale) {  // wp_collision_common.py:11-38
  if (dist > eta) { cost = dist - 0.5f * eta; gscale = 1.0f; }
  else { cost = 0.5f * dist * dist / eta; gscale = dist / eta; }
}

__device__ __forceinline__ f3 quat_rot(float qw, float qx, float qy, float qz, f3 v) {  // warp quat_rotate
  const f3 q = make_f3(qx, qy, qz);
  const float c = 2.0f * qw * qw - 1.0f, d = 2.0f * dot(q, v);
  const f3 cr = cross(q, v);
  return make_f3(v.x * c + q.x * d + cr.x * 2.0f * qw, v.y * c + q.y * d + cr.y * 2.0f * qw, v.z * c + q.z * d + cr.z * 2.0f * qw);
}

// one obstacle slot of a mesh set as the kernels consume it
struct MeshSlot {
  curobo_hip_mesh m;
  f3 t;
  float qw, qx, qy, qz, max_half_diag;
  bool enabled;
};
__device__ __forceinline__ MeshSlot load_mesh_slot(const curobo_hip_mesh_set &ms, int env, int o) {
  MeshSlot s;
  const int flat = env * ms.max_n + o;
  s.enabled = o < ms.count[env] && ms.enable[flat] == 1;  // is_obs_enabled (data_mesh.py:555-575)
  s.m = ms.meshes[s.enabled ? ms.mesh_id[flat] : 0];
  const float *ip = ms.inv_pose + (size_t)flat * 8;  // x y z qw qx qy qz pad: world -> mesh frame
  s.t = make_f3(ip[0], ip[1], ip[2]);
  s.qw = ip[3]; s.qx = ip[4]; s.qy = ip[5]; s.qz = ip[6];
  const float *dm = ms.dims + (size_t)flat * 4;
  s.max_half_diag = 0.5f * sqrtf(dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2]);
  return s;
}
__device__ __forceinline__ f3 mesh_to_local(const MeshSlot &s, f3 v) { return quat_rot(s.qw, s.qx, s.qy, s.qz, v) + s.t; }

// Early reject (result preserving): the surface lies inside the mesh's bounding box (root of the tree), so the signed
// distance of a point outside the box is at least its distance to the box; when that exceeds r_adj + the half sweep
// length (+ rounding) neither the centre nor any sweep sample can penetrate.
__device__ __forceinline__ bool mesh_early_reject(const MeshSlot &s, f3 lc, float r_adj, float reach) {
  const float *rb = s.m.node_box + 8;
  const float ex = fmaxf(fmaxf(rb[0] - lc.x, lc.x - rb[4]), 0.0f), ey = fmaxf(fmaxf(rb[1] - lc.y, lc.y - rb[5]), 0.0f),
              ez = fmaxf(fmaxf(rb[2] - lc.z, lc.z - rb[6]), 0.0f);
  const float thr = r_adj + reach;
  return ex * ex + ey * ey + ez * ez > thr * thr * 1.00001f;
}

// Cost and mesh-frame gradient of ONE sphere against ONE mesh that passed the early reject: the centre sample plus the
// sweep towards the previous / next point (wp_sweep_collision_kernel.py:176-254; the mesh twin of
// scene_device.hpp::obstacle_contribution).  lc = centre in the mesh frame, reach as for mesh_early_reject.
// (QUERY: sdf = query(point, radius that matters, max_distance, may be inside, gradient&) -- mesh_sdf_within for a lane on its own,
// mesh_block_sdf for a workgroup that answers one sphere's queries together)
template <int SWEEP, class QUERY>
__device__ __forceinline__ void mesh_contribution_q(const MeshSlot &s, int gradient_mode, f3 lc, bool has_prev, bool has_next, f3 prev_c,
                                                    f3 next_c, float r_adj, float eta, float half_w_prev, float half_w_next,
                                                    float reach, float &cost_sum, f3 &grad_local, QUERY query) {
  // max_distance = max(half the bounding-box diagonal, the query distance) (data_mesh.py:660-668)
  const float max_distance = fmaxf(s.max_half_diag, r_adj);
  // how far the centre's distance matters: its cost below r_adj, the sweep culling below r_adj + the half segment
  const float cull_slack = 2e-6f + 1e-6f * max_distance;
  // The samples of an item -- the centre, then up to SWEEP - 1 per direction -- are queried by ONE loop (one inlined copy of
  // the walk; in a wavefront the lanes' i-th queries run together).  dir = -1: the centre.
  f3 g_c = make_f3(0.f, 0.f, 0.f), ln = lc, qp = lc;
  float sdf_c = 0.0f, pen_c = 0.0f, c_c = 0.0f, gs_c = 0.0f, half_dist = 0.0f, inv_half = 0.0f, jump = 0.0f;
  float q_radius = (r_adj + reach + cull_slack) * 1.0001f + 1e-6f;
  bool q_may_in = true;
  int dir = -1, k = 0;
#pragma unroll 1
  for (;;) {
    f3 g;
    const float sdf = query(qp, q_radius, max_distance, q_may_in, g);
    if (gradient_mode == 1 && sdf > 0.0f) g = -1.0f * g;
    const float pen = -sdf + r_adj;
    float c = 0.0f, gs = 0.0f;
    if (pen > 0.0f) {
      activation_m(pen, eta, c, gs);
      cost_sum += c;
      grad_local = grad_local + gs * g;
    }
    if (dir < 0) { sdf_c = sdf; pen_c = pen; c_c = c; gs_c = gs; g_c = g; }
    else {  // a sweep sample: the step along the segment (wp_sweep_collision_kernel.py:204-212)
      if (pen > 0.0f) jump += pen;
      else if (-pen >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen, r_adj);
      k++;
    }
    if (SWEEP == 0) break;
    // ---- the next sample to query, if any
    bool have = false;
#pragma unroll 1
    while (!have) {
      if (dir >= 0 && k < SWEEP && !(jump >= half_dist)) { have = true; break; }
      dir++;
      if (dir >= 2) break;
      if (!(dir == 0 ? has_prev : has_next)) continue;
      // sweep culling (result preserving, as for cuboids: scene_device.hpp): every sample lies within the half segment
      // length of the centre and the signed distance is 1-Lipschitz, so a centre that is clear by more than that cannot
      // have a penetrating sample.  (Closed meshes only: under the reference's ray rule an open mesh's sign changes across
      // the shadow lines of its rim, far from any surface -- sign_rule 1 culls nothing by continuity.)  (A centre that found no surface within its search radius is clear by more than any
      // half segment -- or beyond max_distance, where the samples find nothing either.)
      if (s.m.sign_rule == 0 && -pen_c > (dir == 0 ? half_w_prev : half_w_next) * 1.0001f + cull_slack) continue;
      ln = mesh_to_local(s, dir == 0 ? prev_c : next_c);
      const f3 dd = ln - lc;
      half_dist = sqrtf(dot(dd, dd)) * 0.5f;
      inv_half = 1.0f / fmaxf(half_dist, 0.001f);
      jump = 0.0f;
      k = SWEEP;  // (no sample unless the direction is entered below)
      if (jump >= half_dist) continue;
      // k = 0 of the reference's loop samples t = 1, the centre itself: its terms are added again, not walked again
      if (pen_c > 0.0f) { cost_sum += c_c; grad_local = grad_local + gs_c * g_c; jump += pen_c; }
      else if (-pen_c >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen_c, r_adj);
      k = 1;
    }
    if (!have) break;
    const float tt = 1.0f - 0.5f * jump * inv_half;
    qp = tt * lc + (1.0f - tt) * ln;
    // the sample's distance matters below r_adj (cost) and below r_adj + the rest of the half segment (next step);
    // it lies within half_dist of the centre, so it can only be deep inside when the centre is inside
    q_radius = (r_adj + (half_dist - jump)) * 1.0001f + 1e-6f;
    q_may_in = s.m.sign_rule != 0 || sdf_c < half_dist;
  }
}

template <int SWEEP>
__device__ __forceinline__ void mesh_contribution(const MeshSlot &s, int gradient_mode, f3 lc, bool has_prev, bool has_next, f3 prev_c,
                                                  f3 next_c, float r_adj, float eta, float half_w_prev, float half_w_next,
                                                  float reach, float &cost_sum, f3 &grad_local) {
  mesh_contribution_q<SWEEP>(s, gradient_mode, lc, has_prev, has_next, prev_c, next_c, r_adj, eta, half_w_prev, half_w_next, reach, cost_sum,
                             grad_local, [&](f3 qp, float q_radius, float max_distance, bool may_in, f3 &g) {
                               return mesh_sdf_within(s.m, qp, q_radius, max_distance, may_in, g);
                             });
}

// ---- the same contribution by a GROUP of G lanes (sphere_mesh_walk_kernel), every query of the item in ONE loop.
// With the queries as an outer loop around a group walk (a function of its own through most of round 4) the groups of a
// wavefront meet again after every query: the wavefront makes (sum over its query rounds of the LONGEST walk of the round) passes through the loop body --
// 74 on the bench's mesh world where a sphere needs 17 moves.  Here a group that has finished a query settles it and
// starts its next one while the others are still walking; the wavefront's pass count is the longest TOTAL of its groups.
// Per pass every group makes one move (down into the nearest descendant / to the nearest sibling still owed / a leaf's
// triangles) or one transition between queries.
template <int SWEEP, int G>
__device__ __forceinline__ void mesh_contribution_group(const MeshSlot &s, int gradient_mode, f3 lc, bool has_prev, bool has_next, f3 prev_c,
                                                        f3 next_c, float r_adj, float eta, float half_w_prev, float half_w_next,
                                                        float reach, float &cost_sum, f3 &grad_local, float *keys, int key_stride, unsigned stat_q = 0u) {
  constexpr int LV = G == 16 ? 4 : 3;  // tree levels per step
  constexpr unsigned GM = (1u << G) - 1u;
  constexpr float FAR = 3.0e38f;
  const curobo_hip_mesh &m = s.m;
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  const int g = threadIdx.x & (G - 1), gbase = threadIdx.x & (64 - G);
  const int depth_leaves = 31 - __builtin_clz(m.n_leaves);
  const float max_distance = fmaxf(s.max_half_diag, r_adj);      // (mesh_contribution)
  const float cull_slack = 2e-6f + 1e-6f * max_distance;
  // the item's sweep state (mesh_contribution)
  f3 g_c = make_f3(0.f, 0.f, 0.f), ln = lc, qp = lc;
  float sdf_c = 0.0f, pen_c = 0.0f, c_c = 0.0f, gs_c = 0.0f, half_dist = 0.0f, inv_half = 0.0f, jump = 0.0f;
  float q_radius = (r_adj + reach + cull_slack) * 1.0001f + 1e-6f;
  bool q_may_in = true;
  int dir = -1, k = 0;
  // the running query (mesh_sdf_within + the walk): every lane keeps the best of ITS triangles, only the
  // distance is shared while walking (it prunes); which lane holds the closest point is settled when the walk is over
  bool full = !(q_radius < max_distance);
  float best_d2 = 0.0f, limit_d2 = 0.0f, l_d2 = FAR;
  f3 l_c = lc;
  int l_t = 0, l_region = 0, node = 0, gl = 0;
  bool l_tie = false;
  unsigned long long owed = 0ull;
  // The closest point found by an earlier sample of this item bounds the distance of the next one (triangle inequality:
  // |q' - c| <= |q' - q| + |q - c|): the walk of a sweep sample then prunes with the sample's step, not with the search
  // radius -- for a sphere deep inside a mesh (nothing within the radius: every sample is searched at the full range, and
  // the seven samples of such a sphere are the longest chain of the launch) that is centimetres instead of the mesh's
  // half diagonal.  Only the pruning bound shrinks (never below the true distance): the same closest point is found.
  bool have_prev = false;
  float prev_d = 0.0f;
  f3 prev_qp = lc;
  auto begin_query = [&]() {
    if (g == 0) CUROBO_MESH_COUNT(0, 1);
    const float r = full ? max_distance : q_radius;
    limit_d2 = best_d2 = r * r;
    if (MESH_PREV_BOUND && have_prev) {
      const f3 dq = qp - prev_qp;
      const float ub = (prev_d + sqrtf(dot(dq, dq))) * 1.00001f + 1e-6f;
      best_d2 = fminf(best_d2, ub * ub);
    }
    l_d2 = FAR;
    l_tie = false;
    owed = 0ull;
    gl = 0;
    node = box_dist2(box[2], box[3], qp) > best_d2 ? 0 : 1;
  };
  begin_query();
#pragma unroll 1
  for (;;) {
#ifdef CUROBO_MESH_STATS
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(1)) - 1) CUROBO_MESH_COUNT(3, 1);  // passes of the wavefront
    if (g == 0 && node == 0) CUROBO_MESH_COUNT(4, 1);                                               // transitions
    if (g == 0) g_mesh_lane[(stat_q & 0x1ffffu) + (node == 0 ? 0x20000u : 0u)] += 1;               // per item: moves | transitions
#endif
    if (node != 0) {
      // ================= one move of the walk
      const int d0 = min(gl * LV, depth_leaves);
      bool descended = false;
      if (d0 == depth_leaves) {  // a leaf: lane j tests triangles j, j + G, ...
        if (g == 0) CUROBO_MESH_COUNT(2, 1);
        const int t0 = (node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
        for (int t = t0 + g; t < t1; t += G) {
          const TriRec r = tri[t];
          int region;
          const f3 c = closest_on_triangle(qp, make_f3(r.a.x, r.a.y, r.a.z), make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z), region);
          const f3 d = qp - c;
          const float d2 = dot(d, d);
          if (d2 <= l_d2) { l_tie = d2 == l_d2; l_d2 = d2; l_c = c; l_t = t; l_region = region; }
        }
        best_d2 = fminf(best_d2, group_min<G>(l_d2));
      } else {  // an interior node: its 2^sl descendants sl levels down, one per lane
        if (g == 0) CUROBO_MESH_COUNT(1, 1);
        const int sl = min(LV, depth_leaves - d0), cnt = 1 << sl;
        const int child = (node << sl) + (g < cnt ? g : 0);
        const float4 lo = box[child * 2], hi = box[child * 2 + 1];
        const float dist = g < cnt ? box_dist2(lo, hi, qp) : FAR;
        // nearest first; boxes the point is inside of order by the distance to their centres: one key, negative for those
        const f3 cv = make_f3(lo.x + hi.x, lo.y + hi.y, lo.z + hi.z) - 2.0f * qp;
        const float key = dist > 0.0f ? dist : -__frcp_rn(1.0f + dot(cv, cv));
        keys[gl * key_stride] = key;
        const bool in = key <= best_d2;
        const unsigned mask = group_ballot<G>(in);
        if (mask != 0u) {
          const float kmin = group_min<G>(in ? key : FAR);
          const int j0 = __ffs((int)group_ballot<G>(in && key == kmin)) - 1;
          owed = (owed & ~((unsigned long long)GM << (G * gl))) | ((unsigned long long)(mask & ~(1u << j0)) << (G * gl));
          node = (node << sl) + j0;
          ++gl;
          descended = true;
        }
      }
      if (!descended) {  // the nearest sibling still owed a visit and still in range, else up
        bool more = false;
        while (gl > 0) {
          const int lvl = gl - 1, sp = min(LV, depth_leaves - lvl * LV);
          const int parent = node >> sp;
          const unsigned rest = (unsigned)(owed >> (G * lvl)) & GM;
          if (rest != 0u) {
            const float key = keys[lvl * key_stride];
            const bool in = ((rest >> g) & 1u) != 0u && key <= best_d2;
            const unsigned mask = group_ballot<G>(in);
            if (mask != 0u) {
              const float kmin = group_min<G>(in ? key : FAR);
              const int j = __ffs((int)group_ballot<G>(in && key == kmin)) - 1;
              owed = (owed & ~((unsigned long long)GM << (G * lvl))) | ((unsigned long long)(mask & ~(1u << j)) << (G * lvl));
              node = (parent << sp) + j;
              more = true;
              break;
            }
            owed &= ~((unsigned long long)GM << (G * lvl));
          }
          node = parent;
          gl = lvl;
        }
        if (!more) node = 0;
      }
      continue;
    }
    // ================= the walk of a query is over: settle it (mesh_sdf_within)
    const float dmin = group_min<G>(l_d2);
    const bool found = dmin <= limit_d2;
    if (!found && !full && q_may_in) {
      const float *rb = m.node_box + 8;  // root box: a point outside it is outside the (closed) surface
      if (!(qp.x < rb[0] || qp.y < rb[1] || qp.z < rb[2] || qp.x > rb[4] || qp.y > rb[5] || qp.z > rb[6])) {
        full = true;  // inside the box, nothing within the radius: far outside in a concavity -- or deep inside
        begin_query();
        continue;
      }
    }
    float sdf = max_distance;
    f3 gq = make_f3(0.f, 0.f, 0.f);
    if (found) {
      const unsigned win = group_ballot<G>(l_d2 == dmin);
      const int src = gbase | (__ffs((int)win) - 1);
      // (the same distance from several triangles: the closest point lies on a feature they share -- or on two separate
      // ones; a face verdict is then not trusted)
      const bool tie = (__builtin_popcount(win) > 1) || (__shfl((int)l_tie, src, 64) != 0);
      const f3 cp = make_f3(__shfl(l_c.x, src, 64), __shfl(l_c.y, src, 64), __shfl(l_c.z, src, 64));
      const int side = mesh_feature_side(m, qp, cp, dmin, __shfl(l_t, src, 64), __shfl(l_region, src, 64), tie);
      const float d = sqrtf(dmin);
      if (d > 1e-6f) gq = (1.0f / d) * (qp - cp);
      const bool inside = mesh_point_inside(m, qp, side);
      sdf = inside ? -d : d;
      have_prev = true; prev_d = d; prev_qp = qp;
    }
    // ================= its terms, and the next sample (mesh_contribution)
    if (gradient_mode == 1 && sdf > 0.0f) gq = -1.0f * gq;
    const float pen = -sdf + r_adj;
    float c = 0.0f, gs = 0.0f;
    if (pen > 0.0f) {
      activation_m(pen, eta, c, gs);
      cost_sum += c;
      grad_local = grad_local + gs * gq;
    }
    if (dir < 0) { sdf_c = sdf; pen_c = pen; c_c = c; gs_c = gs; g_c = gq; }
    else {
      if (pen > 0.0f) jump += pen;
      else if (-pen >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen, r_adj);
      k++;
    }
    if (SWEEP == 0) break;
    bool have = false;
#pragma unroll 1
    while (!have) {
      if (dir >= 0 && k < SWEEP && !(jump >= half_dist)) { have = true; break; }
      dir++;
      if (dir >= 2) break;
      if (!(dir == 0 ? has_prev : has_next)) continue;
      if (s.m.sign_rule == 0 && -pen_c > (dir == 0 ? half_w_prev : half_w_next) * 1.0001f + cull_slack) continue;
      ln = mesh_to_local(s, dir == 0 ? prev_c : next_c);
      const f3 dd = ln - lc;
      half_dist = sqrtf(dot(dd, dd)) * 0.5f;
      inv_half = 1.0f / fmaxf(half_dist, 0.001f);
      jump = 0.0f;
      k = SWEEP;
      if (jump >= half_dist) continue;
      if (pen_c > 0.0f) { cost_sum += c_c; grad_local = grad_local + gs_c * g_c; jump += pen_c; }
      else if (-pen_c >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen_c, r_adj);
      k = 1;
    }
    if (!have) break;
    const float tt = 1.0f - 0.5f * jump * inv_half;
    qp = tt * lc + (1.0f - tt) * ln;
    q_radius = (r_adj + (half_dist - jump)) * 1.0001f + 1e-6f;
    q_may_in = s.m.sign_rule != 0 || sdf_c < half_dist;
    full = !(q_radius < max_distance);
    begin_query();
  }
}

// ---- distance-sorted closest-triangle cell lists (curobo_hip_mesh.cell_start / cell_list; built by mesh_bvh.hip).
// The tree walk above is a chain of dependent box fetches -- 17 moves per sphere on the bench's mesh world, 129 in the worst
// query, ~1.2 us a move -- and its divergent control flow is a third of its instruction stream.  A query through the cell
// lists has no walk: one fetch for the cell, then the triangles listed for it, in order of their distance from the cell's
// centre c, as far as any of them can matter: the closest triangle t* of a point p at delta = |p - c| satisfies
// dist(c, t*) <= dist(p, t*) + delta <= dist(p, t_c) + delta <= dist(c, t_c) + 2 delta, where t_c is the first entry -- so the
// prefix of the list up to dist(c, t_c) + 2 delta holds it, and once some triangle at distance best from p is known, the
// prefix up to best + delta.  The G lanes of the group fetch U x G entries of that prefix at once, then their triangles at
// once, and tighten it with what they found; the first round usually is the only one.  The closest point is the exact one -- the same minimum over the same fp32
// point-triangle distances as the walk finds.  A list that ends before that prefix does (cells without a list, the single-entry
// lists of cells far outside) and points outside the grid that the radius still reaches go to the walk.
// what the select kernel needs of a mesh's cell grid to drop a sphere the bounding-box test let through: the packed cell words
// carry the distance dc of every cell's centre from the surface and whether the whole cell is outside it
struct MeshGridRec {
  const uint2 *cell_start;  // nullptr: no cells (or a mesh signed by the reference's rays: no continuity to argue with)
  float lx, ly, lz, h;
  int nx, ny, nz;
};
__device__ __forceinline__ MeshGridRec load_grid_rec(const curobo_hip_mesh &m, bool enabled) {
  MeshGridRec g;
  g.cell_start = (enabled && m.sign_rule == 0) ? reinterpret_cast<const uint2 *>(m.cell_start) : nullptr;
  g.lx = m.grid_lo[0]; g.ly = m.grid_lo[1]; g.lz = m.grid_lo[2]; g.h = m.grid_h;
  g.nx = m.grid_n[0]; g.ny = m.grid_n[1]; g.nz = m.grid_n[2];
  return g;
}
// true when a sphere whose centre is lc (mesh frame) cannot touch the surface within `thr` (= r_adj + the reach of its sweep)
// anywhere along its sweep: its cell is wholly outside the (closed) surface and its centre farther than thr from it.  Result
// preserving for the reasons of mesh_early_reject, with the exact distance of the cell's centre in place of the bounding box.
__device__ __forceinline__ bool mesh_cell_clear(const MeshGridRec &g, f3 lc, float thr, int &list_len) {
  list_len = -1;  // unknown: no cells, or outside the grid
  if (g.cell_start == nullptr) return false;
  const float inv_h = __frcp_rn(g.h);
  const float fx = (lc.x - g.lx) * inv_h, fy = (lc.y - g.ly) * inv_h, fz = (lc.z - g.lz) * inv_h;
  const int ix = (int)floorf(fx), iy = (int)floorf(fy), iz = (int)floorf(fz);
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && ix < g.nx && iy < g.ny && iz < g.nz)) return false;
  const int cell = (ix * g.ny + iy) * g.nz + iz;
  const uint2 rec = g.cell_start[cell];
  list_len = (int)((g.cell_start[cell + 1].x & 0x3fffffffu) - (rec.x & 0x3fffffffu)) - 1;
  if ((rec.x >> 30) != 1u) return false;
  const f3 dv = lc - make_f3(g.lx + ((float)ix + 0.5f) * g.h, g.ly + ((float)iy + 0.5f) * g.h, g.lz + ((float)iz + 0.5f) * g.h);
  const float delta = sqrtf(dot(dv, dv)) * 1.0001f + 2e-6f;
  return __int_as_float((int)rec.y) - delta > thr * 1.0001f + 2e-6f;
}

struct CellEntry {  // 16 bytes
  int32_t tri;   // sorted-order triangle index; -1: the sentinel
  float dist;    // distance of the triangle from the cell's centre c (sentinel: the cover radius of the list)
  float ox, oy;  // octahedral code of the unit vector n from the triangle's closest point q towards c.  The triangle lies
                 // behind the plane through q with normal n, so for ANY p: dist(p, t) >= dist + (p - c) . n -- a bound that
                 // knows in which direction p left the centre (the plain triangle inequality only knows how far)
};
__device__ __forceinline__ void oct_encode(f3 n, float &ox, float &oy) {  // n: unit
  const float s = 1.0f / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
  float x = n.x * s, y = n.y * s;
  if (n.z < 0.0f) {
    const float tx = (1.0f - fabsf(y)) * (x >= 0.0f ? 1.0f : -1.0f), ty = (1.0f - fabsf(x)) * (y >= 0.0f ? 1.0f : -1.0f);
    x = tx; y = ty;
  }
  ox = x; oy = y;
}
__device__ __forceinline__ f3 oct_decode(float ox, float oy) {
  f3 n = make_f3(ox, oy, 1.0f - fabsf(ox) - fabsf(oy));
  const float t = fmaxf(-n.z, 0.0f);
  n.x += n.x >= 0.0f ? -t : t;
  n.y += n.y >= 0.0f ? -t : t;
  return __frsqrt_rn(dot(n, n)) * n;
}

// what a query needs of a mesh record, read once per (sphere, mesh) item.  (The lanes of a wavefront query different meshes: the
// whole record by value is 22 vector registers, and reading a field where it is used puts a fetch in front of every query.)
// The grid's numbers live in the group's LDS record (see MESH_ST_*): a query reads them once, at its start.
struct MeshCellsView {
  const curobo_hip_mesh *mp;
  const uint2 *cell_start;
  const CellEntry *list;
  const TriRec *tri;
  const float *st;  // the group's LDS record
};
// ---- the LDS record of a group of lanes working on one sphere (floats; written and read by every lane of the group with the
// same values -- a wavefront's LDS operations execute in order, so no barrier is involved): what the sphere's program needs
// only BETWEEN queries -- the sphere and its neighbours, the slot's pose, the centre sample's terms, the sums.  Measured (notebook,
// round 6): it does NOT lower the kernel's register count -- the compiler forwards the record through registers (142 with it,
// 147 without), and forcing the traffic with `volatile` raised it to 197; the count is set inside closest_on_triangle with a
// round's four entries in flight.  Kept because the item program reads better against named slots than against fifteen arguments.
enum : int {
  MESH_ST_CENTER = 0, MESH_ST_PREV = 3, MESH_ST_NEXT = 6, MESH_ST_HALF_PREV = 9, MESH_ST_HALF_NEXT = 10,
  MESH_ST_T = 11, MESH_ST_Q = 14 /* w x y z */, MESH_ST_GC = 18, MESH_ST_PEN_C = 21, MESH_ST_C_C = 22, MESH_ST_GS_C = 23,
  MESH_ST_DSUM = 24, MESH_ST_GSUM = 25, MESH_ST_COST = 28, MESH_ST_GRAD = 29,
  MESH_ST_GRID_LO = 32, MESH_ST_GRID_H = 35, MESH_ST_GRID_PAD = 36, MESH_ST_GRID_N = 37, MESH_ST_WORDS = 40
};
__device__ __forceinline__ f3 st_load3(const float *st, int i) { return make_f3(st[i], st[i + 1], st[i + 2]); }
__device__ __forceinline__ void st_store3(float *st, int i, f3 v) { st[i] = v.x; st[i + 1] = v.y; st[i + 2] = v.z; }
__device__ __forceinline__ MeshCellsView load_cells_view(const curobo_hip_mesh *mp, float *st) {
  MeshCellsView v;
  v.mp = mp;
  v.cell_start = reinterpret_cast<const uint2 *>(mp->cell_start);
  v.list = reinterpret_cast<const CellEntry *>(mp->cell_list);
  v.tri = reinterpret_cast<const TriRec *>(mp->tri);
  v.st = st;
  st[MESH_ST_GRID_LO] = mp->grid_lo[0]; st[MESH_ST_GRID_LO + 1] = mp->grid_lo[1]; st[MESH_ST_GRID_LO + 2] = mp->grid_lo[2];
  st[MESH_ST_GRID_H] = mp->grid_h; st[MESH_ST_GRID_PAD] = mp->grid_pad;
  st[MESH_ST_GRID_N] = __int_as_float(mp->grid_n[0]); st[MESH_ST_GRID_N + 1] = __int_as_float(mp->grid_n[1]);
  st[MESH_ST_GRID_N + 2] = __int_as_float(mp->grid_n[2]);
  return v;
}

template <int G>
__device__ __forceinline__ float group_bcast0(float v) {  // the value of lane 0 of the group
  return __shfl(v, (int)(threadIdx.x & (64u - G)), 64);
}

// (p - closest point) . unit (pseudo)normal of the feature of triangle t the closest point lies on: positive outside
__device__ __forceinline__ float mesh_side_term(const TriRec *tri, const float *tri_pn, int t, int region, f3 d) {
  f3 n;
  if (region == 0) {
    const TriRec r = tri[t];
    n = cross(make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z));
  } else {
    if (tri_pn == nullptr) return 0.0f;
    const float4 n4 = reinterpret_cast<const float4 *>(tri_pn)[(size_t)t * 6 + (region - 1)];
    n = make_f3(n4.x, n4.y, n4.z);
  }
  const float nn = dot(n, n);
  return nn > 1e-30f ? dot(d, n) * __frsqrt_rn(nn) : 0.0f;
}
template <int G>
__device__ __forceinline__ float group_sum_dpp(float v) {  // G = 4, 8, 16 or 32 aligned lanes; every lane gets the sum
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  if (G >= 8) v += dpp_f<0x141>(v);
  if (G >= 16) v += dpp_f<0x140>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// the signed distance of qp (mesh frame) and the reference's local gradient, by the G lanes of an aligned group that all hold
// the same query.  Returns false when the cell lists cannot answer it (the caller walks the tree instead).  A surface
// farther than `q_radius` from a point OUTSIDE it may be reported as "nothing within max_distance" (max_distance, 0), exactly
// as mesh_sdf_within does -- the callers' results are the same either way (see mesh_sdf_within).
// Answers: MESH_CELLS_OK; MESH_CELLS_TO_WALK -- the lists do not apply (no lists, a mesh signed by rays, a point outside the grid
// that the radius still reaches: the tree walk prunes those well); MESH_CELLS_TO_WIDE -- the lists apply and this group of G lanes
// is the wrong tool: a cell without a list (more candidates than the build gathers), a list that ends before the prefix does, or
// a prefix still open after MESH_CELLS_MAX_ROUNDS rounds (a point about equally far from hundreds of triangles: the centre of a
// ball) -- such a sphere is answered by a whole workgroup (sphere_mesh_wide_kernel), not by eight lanes while the launch waits.
// (Only for meshes of up to MESH_WIDE_MAX_LEAVES leaves = 32 k triangles: the workgroup's scan is over ALL leaves within its bound.
// On a dense mesh a cell has no list because many SMALL triangles are near it, not because they are equally far: the tree prunes
// those, and a scan of 20 k leaf boxes per query would not -- 159 k triangles, 23 k spheres sent there: 2.5 ms either way
// (tools/r06/big_mesh_probe.py); such a mesh keeps the tree walk for what its lists cannot answer.)
enum : int { MESH_CELLS_OK = 0, MESH_CELLS_TO_WALK = 1, MESH_CELLS_TO_WIDE = 2 };
#ifndef MESH_CELLS_MAX_ROUNDS
#define MESH_CELLS_MAX_ROUNDS 16
#endif
#ifndef MESH_WIDE_MAX_LEAVES
#define MESH_WIDE_MAX_LEAVES 4096
#endif
template <int G, int U>
__device__ __forceinline__ int mesh_cells_sdf(const MeshCellsView &mv, f3 qp, float q_radius, float max_distance, float &sdf, f3 &grad,
                                              unsigned stat_q = 0u) {
  constexpr float FAR = 3.0e38f;
  sdf = max_distance;
  grad = make_f3(0.f, 0.f, 0.f);
  if (mv.cell_start == nullptr) return MESH_CELLS_TO_WALK;
  const int g = threadIdx.x & (G - 1), gbase = threadIdx.x & (64 - G);
  const float lx = mv.st[MESH_ST_GRID_LO], ly = mv.st[MESH_ST_GRID_LO + 1], lz = mv.st[MESH_ST_GRID_LO + 2], h = mv.st[MESH_ST_GRID_H];
  const int nx = __float_as_int(mv.st[MESH_ST_GRID_N]), ny = __float_as_int(mv.st[MESH_ST_GRID_N + 1]), nz = __float_as_int(mv.st[MESH_ST_GRID_N + 2]);
  const float inv_h = __frcp_rn(h);
  const float fx = (qp.x - lx) * inv_h, fy = (qp.y - ly) * inv_h, fz = (qp.z - lz) * inv_h;
  const int ix = (int)floorf(fx), iy = (int)floorf(fy), iz = (int)floorf(fz);
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && ix < nx && iy < ny && iz < nz)) {
    // outside the grid = farther than grid_pad from the bounding box, hence from the surface, and outside it under either
    // sign rule (the reference's rays all have to hit: the point would lie inside the box)
    return q_radius <= mv.st[MESH_ST_GRID_PAD] ? MESH_CELLS_OK : MESH_CELLS_TO_WALK;
  }
  const int cell = (ix * ny + iy) * nz + iz;
  const uint2 rec = mv.cell_start[cell];
  const uint32_t w0 = rec.x, w1 = mv.cell_start[cell + 1].x;
  const int n = (int)((w1 & 0x3fffffffu) - (w0 & 0x3fffffffu));  // >= 1: the sentinel
  const unsigned cell_side = w0 >> 30;
  const float dc = __int_as_float((int)rec.y);  // distance of the cell's centre from the surface
  const CellEntry *list = mv.list + (w0 & 0x3fffffffu);
  const f3 dv = qp - make_f3(lx + ((float)ix + 0.5f) * h, ly + ((float)iy + 0.5f) * h, lz + ((float)iz + 0.5f) * h);
  const float delta = sqrtf(dot(dv, dv)) * 1.0001f + 2e-6f;  // |p - c|, rounded up past the fp32 error of the listed distances
#ifdef CUROBO_MESH_STATS
  if (g == 0) g_mesh_lane[0x20000u + (stat_q & 0x1ffffu)] += 1;  // queries of the queue entry
#endif
  if (g == 0) CUROBO_MESH_COUNT(0, 1);
  // the surface is at least dc - delta away from p: outside and beyond the radius needs no triangle
  if (cell_side == 1u && dc - delta > q_radius) { if (g == 0) CUROBO_MESH_COUNT(4, 1); return MESH_CELLS_OK; }
  if (mv.mp->sign_rule != 0) return MESH_CELLS_TO_WALK;  // (a mesh signed by the reference's rays: see below)
  const int beyond = mv.mp->n_leaves <= MESH_WIDE_MAX_LEAVES ? MESH_CELLS_TO_WIDE : MESH_CELLS_TO_WALK;
  if (n <= 1) return beyond;  // a cell without a list
  // the closest triangle of p is no farther than dc + 2 delta from the centre
  float l_d2 = FAR, lim = dc * 1.000001f + 2.0f * delta, best_d2 = FAR;
  f3 l_c = qp;
  int l_t = 0, l_region = 0;
  float l_tie_term = 0.0f;  // side terms of earlier triangles of this lane at the SAME distance as the kept one
  bool complete = false;
  const float4 *list4 = reinterpret_cast<const float4 *>(list);
  const float dir_slack = delta * 2e-4f + 2e-6f;  // rounding of the decoded direction and of the listed distance
#pragma unroll 1
  for (int k = 0;; k += U * G) {
#ifdef CUROBO_MESH_STATS
    if (g == 0) g_mesh_lane[stat_q & 0x1ffffu] += 1;  // rounds of the queue entry
#endif
    if (g == 0) CUROBO_MESH_COUNT(1, 1);
    // a round: U x G entries, fetched at once (16 bytes each); an entry needs its triangle only when neither its distance
    // nor its directional bound clears it -- after the first few of the first round that is rare, so the rounds of a long
    // prefix (a point deep inside a block: every face is about equally far) are scans
    float4 e[U];
#pragma unroll
    for (int u = 0; u < U; u++) e[u] = list4[min(k + u * G + g, n - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int t = __float_as_int(e[u].x);
      const float dist = e[u].y;
      // dist(p, t) >= dist + (p - c) . n (see CellEntry); a triangle the centre touches has no direction: the plain bound
      const float lb = dist > 1e-6f ? dist + dot(dv, oct_decode(e[u].z, e[u].w)) - dir_slack : dist - delta;
      if (t >= 0 && dist <= lim) CUROBO_MESH_COUNT(5, 1);
      if (t >= 0 && dist <= lim && (lb <= 0.0f || lb * lb <= fminf(best_d2, l_d2))) {
        CUROBO_MESH_COUNT(2, 1);
        const TriRec r = mv.tri[t];
        int region;
        const f3 c = closest_on_triangle(qp, make_f3(r.a.x, r.a.y, r.a.z), make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z), region);
        const f3 d = qp - c;
        const float d2 = dot(d, d);
        if (d2 <= l_d2) {
          // (an exact tie: the triangle kept so far stays in the verdict through its side term)
          l_tie_term = d2 == l_d2 ? l_tie_term + mesh_side_term(mv.tri, mv.mp->tri_pn, l_t, l_region, qp - l_c) : 0.0f;
          l_d2 = d2; l_c = c; l_t = t; l_region = region;
        }
      }
      // what the group found so far clears later entries: no triangle whose bound exceeds it can be closer to p
      if (u < 2 || u == U - 1) best_d2 = fminf(best_d2, group_min<G>(l_d2));
    }
    if (best_d2 < FAR) lim = fminf(lim, sqrtf(best_d2) * 1.000001f + delta);
    // the last entry of the round (uniform over the group): past the prefix -> done; the sentinel -> the list is over, and it
    // was complete only up to its cover
    const int last_tri = __shfl(__float_as_int(e[U - 1].x), gbase | (G - 1), 64);
    const float last_dist = __shfl(e[U - 1].y, gbase | (G - 1), 64);
    if (last_tri < 0) { complete = last_dist >= lim; break; }
    if (last_dist > lim) { complete = true; break; }
    if (k + U * G >= MESH_CELLS_MAX_ROUNDS * U * G) break;  // (uniform over the group)
  }
  if (!complete) return beyond;
  if (!(best_d2 <= max_distance * max_distance)) return MESH_CELLS_OK;  // nothing within max_distance: (max_distance, 0)
  const float best_d = sqrtf(best_d2);
  const bool winner = l_d2 == best_d2;
  const unsigned win = group_ballot<G>(winner);
  const int src = gbase | (__ffs((int)win) - 1);
  const f3 cp = make_f3(__shfl(l_c.x, src, 64), __shfl(l_c.y, src, 64), __shfl(l_c.z, src, 64));
  bool inside;
  if (cell_side != 0u) inside = cell_side == 2u;  // (sign_rule 0 only: the build leaves 0 otherwise)
  else {
    // The side by the closest feature -- the face's normal, or the pseudonormal of the edge / vertex (Baerentzen & Aanaes) -- of
    // EVERY triangle that attains the minimum: the sign of the sum of (p - closest point) . unit normal over them.  One
    // triangle: mesh_feature_side's verdict.  A closest point on an edge that each of its two triangles files under "face" by a
    // rounding: the sum of the two face normals IS the edge's pseudonormal.  A point on the bisector of two faces: both terms
    // agree.  So no query of a closed mesh needs a ray here (the tree walk casts rays where mesh_feature_side gives no verdict:
    // as a call from this kernel that cast cost a fifth of its registers, and handed to the walk kernel one such sphere costs
    // the launch 100 us).  A mesh signed by the reference's rays (sign_rule 1) does go to the tree walk.
    if (mv.mp->sign_rule != 0) return MESH_CELLS_TO_WALK;
    float term = 0.0f;
    if (winner && best_d2 > 1e-12f) term = l_tie_term + mesh_side_term(mv.tri, mv.mp->tri_pn, l_t, l_region, qp - l_c);
    inside = group_sum_dpp<G>(term) < 0.0f;
  }
  if (best_d > 1e-6f) grad = (1.0f / best_d) * (qp - cp);
  sdf = inside ? -best_d : best_d;
  return MESH_CELLS_OK;
}

// one obstacle slot of a mesh set for the cell-list kernel: the mesh record stays in memory (see MeshCellsView), the pose goes
// to the group's LDS record
struct MeshPoseSlot {
  const curobo_hip_mesh *mp;
  float max_half_diag;
  bool enabled;
};
__device__ __forceinline__ MeshPoseSlot load_mesh_pose_slot(const curobo_hip_mesh_set &ms, int env, int o, float *st) {
  MeshPoseSlot s;
  const int flat = env * ms.max_n + o;
  s.enabled = o < ms.count[env] && ms.enable[flat] == 1;
  s.mp = ms.meshes + (s.enabled ? ms.mesh_id[flat] : 0);
  const float4 *ip = reinterpret_cast<const float4 *>(ms.inv_pose + (size_t)flat * 8);
  const float4 p0 = ip[0], p1 = ip[1];
  st[MESH_ST_T] = p0.x; st[MESH_ST_T + 1] = p0.y; st[MESH_ST_T + 2] = p0.z;
  st[MESH_ST_Q] = p0.w; st[MESH_ST_Q + 1] = p1.x; st[MESH_ST_Q + 2] = p1.y; st[MESH_ST_Q + 3] = p1.z;
  const float4 dm = *reinterpret_cast<const float4 *>(ms.dims + (size_t)flat * 4);
  s.max_half_diag = 0.5f * sqrtf(dm.x * dm.x + dm.y * dm.y + dm.z * dm.z);
  return s;
}
__device__ __forceinline__ f3 mesh_to_local_st(const float *st, f3 v) {
  return quat_rot(st[MESH_ST_Q], st[MESH_ST_Q + 1], st[MESH_ST_Q + 2], st[MESH_ST_Q + 3], v) + st_load3(st, MESH_ST_T);
}
__device__ __forceinline__ f3 mesh_to_world_vector_st(const float *st, f3 v) {
  return quat_rot(st[MESH_ST_Q], -st[MESH_ST_Q + 1], -st[MESH_ST_Q + 2], -st[MESH_ST_Q + 3], v);
}

// mesh_contribution with every query through the cell lists, by a group of G lanes; the item's cost and mesh-frame gradient are
// ADDED to st[MESH_ST_COST] / st[MESH_ST_GRAD..].  The sphere (centre, neighbours, half steps) and the slot's pose are read from
// the group's LDS record.  flags: bit 0 / 1 = the previous / next point exists.  Returns false when a query of the item could
// not be answered by the lists: the caller then hands the sphere to the tree walk.
template <int SWEEP, int G, int U>
__device__ __forceinline__ int mesh_contribution_cells(const MeshPoseSlot &s, int gradient_mode, float *st, unsigned flags,
                                                       float r_adj, float eta, float reach, unsigned stat_q = 0u) {
  const float max_distance = fmaxf(s.max_half_diag, r_adj);
  const float cull_slack = 2e-6f + 1e-6f * max_distance;
  const bool lipschitz = s.mp->sign_rule == 0;
  const MeshCellsView mv = load_cells_view(s.mp, st);
  const f3 lc = mesh_to_local_st(st, st_load3(st, MESH_ST_CENTER));
  f3 ln = lc, qp = lc;
  float half_dist = 0.0f, inv_half = 0.0f, jump = 0.0f;
  float q_radius = (r_adj + reach + cull_slack) * 1.0001f + 1e-6f;
  int dir = -1, k = 0;
#pragma unroll 1
  for (;;) {
    f3 g;
    float sdf;
    if (const int code = mesh_cells_sdf<G, U>(mv, qp, q_radius, max_distance, sdf, g, stat_q)) return code;
    if (gradient_mode == 1 && sdf > 0.0f) g = -1.0f * g;
    const float pen = -sdf + r_adj;
    float c = 0.0f, gs = 0.0f;
    if (pen > 0.0f) {
      activation_m(pen, eta, c, gs);
      st[MESH_ST_COST] += c;
      st_store3(st, MESH_ST_GRAD, st_load3(st, MESH_ST_GRAD) + gs * g);
    }
    if (dir < 0) { st[MESH_ST_PEN_C] = pen; st[MESH_ST_C_C] = c; st[MESH_ST_GS_C] = gs; st_store3(st, MESH_ST_GC, g); }
    else {
      if (pen > 0.0f) jump += pen;
      else if (-pen >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen, r_adj);
      k++;
    }
    if (SWEEP == 0) break;
    bool have = false;
#pragma unroll 1
    while (!have) {
      if (dir >= 0 && k < SWEEP && !(jump >= half_dist)) { have = true; break; }
      dir++;
      if (dir >= 2) break;
      if (!((flags >> dir) & 1u)) continue;
      const float pen_c = st[MESH_ST_PEN_C];
      if (lipschitz && -pen_c > st[MESH_ST_HALF_PREV + dir] * 1.0001f + cull_slack) continue;
      ln = mesh_to_local_st(st, st_load3(st, dir == 0 ? MESH_ST_PREV : MESH_ST_NEXT));
      const f3 dd = ln - lc;
      half_dist = sqrtf(dot(dd, dd)) * 0.5f;
      inv_half = 1.0f / fmaxf(half_dist, 0.001f);
      jump = 0.0f;
      k = SWEEP;
      if (jump >= half_dist) continue;
      if (pen_c > 0.0f) {
        st[MESH_ST_COST] += st[MESH_ST_C_C];
        st_store3(st, MESH_ST_GRAD, st_load3(st, MESH_ST_GRAD) + st[MESH_ST_GS_C] * st_load3(st, MESH_ST_GC));
        jump += pen_c;
      }
      else if (-pen_c >= 1000.0f) jump += r_adj;
      else jump += fmaxf(-pen_c, r_adj);
      k = 1;
    }
    if (!have) break;
    const float tt = 1.0f - 0.5f * jump * inv_half;
    qp = tt * lc + (1.0f - tt) * ln;
    q_radius = (r_adj + (half_dist - jump)) * 1.0001f + 1e-6f;
  }
  return MESH_CELLS_OK;
}

// ---- one query answered by a WHOLE WORKGROUP (sphere_mesh_wide_kernel): every thread calls with the same arguments and gets the
// same result.  The exact closest point over the leaves whose boxes lie within an a-priori bound -- the distance of the cell's
// centre from the surface + the point's distance from that centre where the mesh has a cell grid, else max_distance -- thread i
// taking leaves i, i + 256, ...: no tree, no list, no order to get wrong; for the points that are sent here (about equally far
// from very many triangles) nothing prunes anyway and 256 lanes are 32 times the eight of the other kernels.  The sign as in
// mesh_cells_sdf (the sum of the side terms of every triangle at the minimum; sign_rule 1: the reference's rays, every lane the
// same walk).  Ties between triangles are settled by the lower triangle index (deterministic).
struct MeshWideLds {
  float w_d2[4], w_term[4], cp[3];
  int w_t[4];
};
__device__ __forceinline__ float mesh_block_sdf(const curobo_hip_mesh &m, f3 qp, float max_distance, f3 &grad, MeshWideLds &L) {
  constexpr float FAR = 3.0e38f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  grad = make_f3(0.f, 0.f, 0.f);
  float ub = max_distance * 1.000001f + 1e-6f;
  if (m.cell_start != nullptr) {
    const float inv_h = __frcp_rn(m.grid_h);
    const float fx = (qp.x - m.grid_lo[0]) * inv_h, fy = (qp.y - m.grid_lo[1]) * inv_h, fz = (qp.z - m.grid_lo[2]) * inv_h;
    const int ix = (int)floorf(fx), iy = (int)floorf(fy), iz = (int)floorf(fz);
    if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && ix < m.grid_n[0] && iy < m.grid_n[1] && iz < m.grid_n[2]) {
      const uint2 rec = reinterpret_cast<const uint2 *>(m.cell_start)[(ix * m.grid_n[1] + iy) * m.grid_n[2] + iz];
      const f3 dv = qp - make_f3(m.grid_lo[0] + ((float)ix + 0.5f) * m.grid_h, m.grid_lo[1] + ((float)iy + 0.5f) * m.grid_h,
                                 m.grid_lo[2] + ((float)iz + 0.5f) * m.grid_h);
      ub = fminf(ub, __int_as_float((int)rec.y) * 1.00001f + sqrtf(dot(dv, dv)) * 1.0001f + 4e-6f);
    }
  }
  const float ub2 = ub * ub;
  float l_d2 = FAR, l_tie_term = 0.0f;
  f3 l_c = qp;
  int l_t = 0x7fffffff, l_region = 0;
  const int used = (m.n_tri + m.leaf_size - 1) / m.leaf_size;
#pragma unroll 1
  for (int leaf = tid; leaf < used; leaf += 256) {
    if (box_dist2(box[(size_t)(m.n_leaves + leaf) * 2], box[(size_t)(m.n_leaves + leaf) * 2 + 1], qp) > ub2) continue;
    const int t1 = min((leaf + 1) * m.leaf_size, m.n_tri);
#pragma unroll 1
    for (int t = leaf * m.leaf_size; t < t1; t++) {
      const TriRec r = tri[t];
      int region;
      const f3 c = closest_on_triangle(qp, make_f3(r.a.x, r.a.y, r.a.z), make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z), region);
      const f3 d = qp - c;
      const float d2 = dot(d, d);
      if (d2 <= l_d2) {  // (triangles come in ascending order within a lane: on a tie the lower index stays the kept one)
        if (d2 == l_d2) l_tie_term += mesh_side_term(tri, m.tri_pn, t, region, d);
        else { l_tie_term = 0.0f; l_d2 = d2; l_c = c; l_t = t; l_region = region; }
      }
    }
  }
  float wmin = l_d2;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wmin = fminf(wmin, __shfl_xor(wmin, o, 64));
  __syncthreads();  // (the previous query's readers are done with L)
  if (lane == 0) L.w_d2[wave] = wmin;
  __syncthreads();
  const float dmin = fminf(fminf(L.w_d2[0], L.w_d2[1]), fminf(L.w_d2[2], L.w_d2[3]));
  if (!(dmin <= max_distance * max_distance)) return max_distance;  // (uniform) nothing within max_distance: (max_distance, 0)
  const bool winner = l_d2 == dmin;
  int wt = winner ? l_t : 0x7fffffff;
  float term = (winner && dmin > 1e-12f) ? l_tie_term + mesh_side_term(tri, m.tri_pn, l_t, l_region, qp - l_c) : 0.0f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    wt = min(wt, __shfl_xor(wt, o, 64));
    term += __shfl_xor(term, o, 64);
  }
  if (lane == 0) { L.w_t[wave] = wt; L.w_term[wave] = term; }
  __syncthreads();
  const int best_t = min(min(L.w_t[0], L.w_t[1]), min(L.w_t[2], L.w_t[3]));
  if (winner && l_t == best_t) { L.cp[0] = l_c.x; L.cp[1] = l_c.y; L.cp[2] = l_c.z; }
  __syncthreads();
  const f3 cp = make_f3(L.cp[0], L.cp[1], L.cp[2]);
  const bool inside = m.sign_rule == 0 ? (L.w_term[0] + L.w_term[1]) + (L.w_term[2] + L.w_term[3]) < 0.0f : mesh_inside_warp_rays(m, qp);
  const float best_d = sqrtf(dmin);
  if (best_d > 1e-6f) grad = (1.0f / best_d) * (qp - cp);
  return inside ? -best_d : best_d;
}

__device__ __forceinline__ f3 mesh_to_world_vector(const MeshSlot &s, f3 v) {  // transform_vector(transform_inverse(inv_t), .)
  return quat_rot(s.qw, -s.qx, -s.qy, -s.qz, v);
}

}  // namespace curobo_hip
