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
// a leaf is one contiguous run.  The query (mesh_device.hpp) is a per-lane STACKLESS traversal -- the heap index is the
// path; nearer child first, prune by the best squared distance so far --; the sign is the parity of ray crossings,
// majority of three rays through the same BVH (closed meshes; the oracle uses the generalised winding number instead:
// two independent methods that must agree).  The collision launch packs the (sphere, mesh) items that survive the
// bounding-box reject and searches each sample only as far as its distance can matter (mesh_sdf_within).
#include <algorithm>

#include "mesh_device.hpp"

#include <hip/hip_fp16.h>

namespace curobo_hip {

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
  const bool inside = mesh_point_inside(a.m, p, found ? side : 0);
  a.out[v] = __float2half(inside ? -d : d);
}

// ------------------------------------------------------------------------------------------------ cell lists (build)
// One lane per cell of the grid (curobo_hip_mesh.cell_start / cell_list; the query is mesh_device.hpp::mesh_cells_sdf).
// Pass 1: the distance dc of the cell's centre c from the surface (the tree walk) and the side c is on; the list has to hold
// every triangle that can be the closest one of SOME point of the cell: for p in the cell, dist(p, t*) <= dist(p, t_c) <=
// dc + hd (hd = half the cell's diagonal), so dist(c, t*) <= dc + 2 hd =: R -- the triangles within R of c are counted by a
// pre-order walk pruned with R.  A cell that is outside the surface and farther from it than the grid's pad can never be
// asked for more than "how far at least" (queries there reach no farther than the pad, or go to the tree walk): it lists
// its nearest triangle only.  A cell whose ball holds more than `gather_cap` triangles (the middle of a blob: everything is
// about equally far) gets no list: its queries walk the tree.  Pass 2 (after the caller's prefix sum): the same walk writes
// (triangle, distance) and a sort key cell << 32 | distance bits; the caller sorts the keys (torch.sort: plumbing) and
// gathers the entries: every list ascending in distance, its sentinel (-1, cover = R) last.
struct MeshCellsArgs {
  curobo_hip_mesh m;       // grid_lo / grid_h / grid_n / grid_pad set, cell pointers unused
  int32_t *count;          // [n_cells]: entries of the cell's list including the sentinel
  float *cover;            // [n_cells]: R of the cell (0: no list)
  uint8_t *side;           // [n_cells]: 1 / 2 = the whole cell is outside / inside, 0 = it may straddle the surface
  float *centre_dist;      // [n_cells]: distance of the cell's centre from the surface
  const int64_t *offsets;  // pass 2: [n_cells + 1] exclusive prefix sum of count
  int64_t *keys;           // pass 2: [offsets[n_cells]]
  int32_t *entries;        // pass 2: [offsets[n_cells]][4] (CellEntry)
  int n_cells, gather_cap;
};

__device__ __forceinline__ f3 mesh_cell_centre(const curobo_hip_mesh &m, int cell) {
  const int iz = cell % m.grid_n[2], iy = (cell / m.grid_n[2]) % m.grid_n[1], ix = cell / (m.grid_n[2] * m.grid_n[1]);
  return make_f3(m.grid_lo[0] + ((float)ix + 0.5f) * m.grid_h, m.grid_lo[1] + ((float)iy + 0.5f) * m.grid_h,
                 m.grid_lo[2] + ((float)iz + 0.5f) * m.grid_h);
}

// every triangle within sqrt(r2) of c, in pre-order of the tree: f(triangle index, distance, c - its closest point)
template <class F>
__device__ __forceinline__ void mesh_ball_walk(const curobo_hip_mesh &m, f3 c, float r2, F f) {
  const float4 *box = reinterpret_cast<const float4 *>(m.node_box);
  const TriRec *tri = reinterpret_cast<const TriRec *>(m.tri);
  unsigned node = 1u;
  while (node != 0u) {
    const bool hit = box_dist2(box[node * 2], box[node * 2 + 1], c) <= r2;  // (an empty padding box is infinitely far)
    if (hit && node < (unsigned)m.n_leaves) { node = node * 2u; continue; }
    if (hit) {
      const int t0 = ((int)node - m.n_leaves) * m.leaf_size, t1 = min(t0 + m.leaf_size, m.n_tri);
      for (int t = t0; t < t1; t++) {
        const TriRec r = tri[t];
        int region;
        const f3 q = closest_on_triangle(c, make_f3(r.a.x, r.a.y, r.a.z), make_f3(r.ab.x, r.ab.y, r.ab.z), make_f3(r.ac.x, r.ac.y, r.ac.z), region);
        const f3 d = c - q;
        const float d2 = dot(d, d);
        if (d2 <= r2) f(t, sqrtf(d2), d);
      }
    }
    node >>= __builtin_ctz(~node);
    node = node ? (node | 1u) : 0u;
  }
}

__global__ void __launch_bounds__(256) mesh_cells_count_kernel(const MeshCellsArgs a) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= a.n_cells) return;
  const curobo_hip_mesh &m = a.m;
  const f3 c = mesh_cell_centre(m, cell);
  const float hd = 0.8660254f * m.grid_h * 1.0001f;
  float d2 = 3.0e38f;
  f3 cp = c;
  int fside = 0;
  mesh_closest_point(m, c, d2, cp, fside);  // (a mesh has a triangle: always found)
  const float dc = sqrtf(d2);
  // the side of the whole cell: every point of it is within hd of c, so a centre farther than hd from the surface has the
  // cell on its side (closed meshes; the reference's ray rule has no such continuity: sign_rule 1 keeps 0)
  unsigned side = 0u;
  if (m.sign_rule == 0 && dc > hd + 1e-6f) side = mesh_point_inside(m, c, fside) ? 2u : 1u;
  const bool far_out = side == 1u && dc - hd > m.grid_pad;
  // (the query's prefix ends at dc (1 + 1e-6) + 2 delta with delta <= hd + 2e-6: the cover has to reach past that for every
  // point of the cell, corners included -- a list that ends a rounding short sends its sphere to the tree walk)
  const float R = far_out ? dc * 1.000001f + 1e-7f : (dc + 2.0f * hd) * 1.00002f + 1e-5f;
  int n = 0;
  mesh_ball_walk(m, c, R * R, [&](int, float, f3) { n++; });
  const bool listed = n <= a.gather_cap;
  a.count[cell] = (listed ? n : 0) + 1;
  a.cover[cell] = listed ? R : 0.0f;
  a.side[cell] = (uint8_t)side;
  a.centre_dist[cell] = dc;
}

__global__ void __launch_bounds__(256) mesh_cells_fill_kernel(const MeshCellsArgs a) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= a.n_cells) return;
  const curobo_hip_mesh &m = a.m;
  const int64_t base = a.offsets[cell];
  const int n = (int)(a.offsets[cell + 1] - base) - 1;
  const float R = a.cover[cell];
  if (n > 0) {
    int i = 0;
    mesh_ball_walk(m, mesh_cell_centre(m, cell), R * R, [&](int t, float d, f3 v) {
      if (i < n) {  // (the count pass walked the same nodes: i never reaches n)
        float ox = 0.0f, oy = 0.0f;
        if (d > 1e-6f) oct_encode((1.0f / d) * v, ox, oy);
        reinterpret_cast<float4 *>(a.entries)[base + i] = make_float4(__int_as_float(t), d, ox, oy);
        a.keys[base + i] = ((int64_t)cell << 32) | (int64_t)__float_as_int(d);
      }
      i++;
    });
  }
  reinterpret_cast<float4 *>(a.entries)[base + n] = make_float4(__int_as_float(-1), R, 0.0f, 0.0f);
  a.keys[base + n] = ((int64_t)cell << 32) | 0x7fffffffll;
}

__global__ void __launch_bounds__(256) mesh_cells_start_kernel(uint2 *cell_start, const int64_t *offsets, const uint8_t *side,
                                                               const float *centre_dist, int n_cells) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell > n_cells) return;
  cell_start[cell] = make_uint2((uint32_t)offsets[cell] | (cell < n_cells ? (uint32_t)side[cell] << 30 : 0u),
                                cell < n_cells ? (uint32_t)__float_as_int(centre_dist[cell]) : 0u);
}

// ------------------------------------------------------------------------------------------------ sphere vs meshes
struct MeshCollArgs {
  float *distance, *gradient;
  const float *spheres;
  curobo_hip_mesh_set set;
  const float *weight, *eta, *speed_dt;
  const int32_t *env_query_idx;
  int batch, horizon, nspheres, use_multi_env, enable_speed_metric, accumulate;
  int slot0, nslots;  // the obstacle slots [slot0, slot0 + nslots) of every environment are handled by this launch (<= 32)
};

constexpr int kMeshSlotsPerLaunch = 32;

// wp_speed_metric.py:38-93 on this kind's share (the map is linear in (cost, gradient))
__device__ __forceinline__ void mesh_speed_metric(f3 center, f3 pp, f3 np, float dt, float &dsum, f3 &gsum) {
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

// The mesh share of the scene cost, packed like scene_collision_packed_kernel: every lane first runs the cheap part for
// ITS sphere -- transform into each mesh's frame + the result-preserving bounding-box reject -- , the workgroup compacts
// the surviving (sphere, mesh slot) items into an LDS queue (sphere-major, slot ascending) and evaluates the queue
// densely, 256 items per round (one tree walk per lane, whichever sphere it belongs to); every sphere then adds ITS items
// in queue order, i.e. in slot order: the same sums as a per-sphere loop over the slots, no atomics.  With one sphere per
// lane and the slot loop in-lane (the first version) a wavefront walked trees as long as ANY of its 64 spheres was near
// any mesh, one sample after the other: 1 859 us per C2-size batch.
template <int SWEEP>
__global__ void __launch_bounds__(256) sphere_mesh_collision_kernel(const MeshCollArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 256;
  const long total = (long)a.batch * a.horizon * a.nspheres;
  const long sidx0 = (long)blockIdx.x * NT;
  const int tid = threadIdx.x, lane64 = tid & 63, wave = tid >> 6;
  const long sidx = sidx0 + tid;
  const int hs = a.horizon * a.nspheres;
  // LDS: sphere stash [3][256] float4 | results [256] float4 | prefix [256 + 8] int | queue [256 * nslots] u16
  float4 *stash = reinterpret_cast<float4 *>(smem);
  float4 *res = stash + 3 * NT;
  int *prefix = reinterpret_cast<int *>(res + NT);
  uint16_t *queue = reinterpret_cast<uint16_t *>(prefix + NT + 8);
  const bool in = sidx < total;
  const int b = in ? (int)(sidx / hs) : 0;
  const int h = in ? (int)((sidx - (long)b * hs) / a.nspheres) : 0;
  const int env = (in && a.use_multi_env) ? a.env_query_idx[b] : 0;
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;  // neighbours feed the sweep and the speed metric
  const bool nb_prev = in && need_nb && h > 0, nb_next = in && need_nb && h < a.horizon - 1;
  float4 s = make_float4(0.f, 0.f, 0.f, -1.f), ps, ns;
  if (in) s = sph[sidx];
  ps = ns = s;
  if (nb_prev) ps = sph[sidx - a.nspheres];
  if (nb_next) ns = sph[sidx + a.nspheres];
  stash[tid] = s; stash[NT + tid] = ps; stash[2 * NT + tid] = ns;
  const float w = a.weight[0], eta = a.eta[0];
  const curobo_hip_mesh_set &ms = a.set;
  // ---- phase 1: which slots survive the bounding-box reject for my sphere
  uint32_t live = 0u;
  if (in && s.w >= 0.0f) {
    const f3 center = make_f3(s.x, s.y, s.z);
    const float r_adj = s.w + eta;
    float half_w_prev = 0.0f, half_w_next = 0.0f;
    if (SWEEP > 0) {
      if (nb_prev) { const f3 dd = make_f3(ps.x, ps.y, ps.z) - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (nb_next) { const f3 dd = make_f3(ns.x, ns.y, ns.z) - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
    for (int k = 0; k < a.nslots; k++) {
      const MeshSlot slot = load_mesh_slot(ms, env, a.slot0 + k);
      if (!slot.enabled) continue;
      if (!mesh_early_reject(slot, mesh_to_local(slot, center), r_adj, reach)) live |= 1u << k;
    }
  }
  // ---- exclusive prefix of the item counts over the workgroup
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
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      queue[at++] = (uint16_t)(tid | (k << 8));
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
      CUROBO_MESH_COUNT(6, 1);
      const unsigned e = queue[q];
      const int owner = (int)(e & 255u), k = (int)(e >> 8);
      const long osidx = sidx0 + owner;
      const int ob = (int)(osidx / hs);
      const int oh = (int)((osidx - (long)ob * hs) / a.nspheres);
      const int oenv = a.use_multi_env ? a.env_query_idx[ob] : 0;
      const float4 cs = stash[owner], cp = stash[NT + owner], cn = stash[2 * NT + owner];
      const MeshSlot slot = load_mesh_slot(ms, oenv, a.slot0 + k);
      const f3 center = make_f3(cs.x, cs.y, cs.z), pp = make_f3(cp.x, cp.y, cp.z), np = make_f3(cn.x, cn.y, cn.z);
      const bool hp = SWEEP > 0 && oh > 0, hn = SWEEP > 0 && oh < a.horizon - 1;
      const float r_adj = cs.w + eta;
      float half_w_prev = 0.0f, half_w_next = 0.0f;
      if (SWEEP > 0) {
        if (hp) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
        if (hn) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
      }
      const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      mesh_contribution<SWEEP>(slot, ms.gradient_mode, mesh_to_local(slot, center), hp, hn, pp, np, r_adj, eta, half_w_prev,
                               half_w_next, reach, cost_sum, grad_local);
      if (cost_sum > 0.0f) {
        const f3 gw = mesh_to_world_vector(slot, grad_local);
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
  // the speed metric is linear in (cost, gradient): scaling this kind's share on its own and adding it to what the scene
  // launch wrote (accumulate) gives the same sums as the reference's one scaling pass over all kinds (wp_autograd.py:
  // 213-231) -- except for its `cost > 0` guard, which the sum passes whenever a share does
  if (a.enable_speed_metric && nb_prev && nb_next && dsum > 0.0f)
    mesh_speed_metric(make_f3(s.x, s.y, s.z), make_f3(ps.x, ps.y, ps.z), make_f3(ns.x, ns.y, ns.z), a.speed_dt[0], dsum, gsum);
  float4 *grad = reinterpret_cast<float4 *>(a.gradient);
  if (a.accumulate) {
    if (dsum > 0.0f) {
      a.distance[sidx] += dsum;
      const float4 g0 = grad[sidx];
      grad[sidx] = make_float4(g0.x + gsum.x, g0.y + gsum.y, g0.z + gsum.z, g0.w);
    }
    return;
  }
  a.distance[sidx] = dsum;
  grad[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
}

// The same launch split in two (curobo_hip_sphere_mesh_collision_ws): spheres that survive the bounding-box reject for ANY
// slot are rare -- 0.06 of them on the C2 shapes -- and sit in a few places of the batch (the links near an obstacle), so
// even the workgroup-level packing above leaves most workgroups idle while a few walk trees for hundreds of
// microseconds.  (1) select: the cheap part for every sphere; survivors go to ONE queue of the launch (a workgroup
// reserves its run with one atomic: order inside a run = sphere order), the others get their zeros; (2) walk: the queue
// evaluated densely by a grid sized for the chip, one live sphere per lane, its slots in ascending order in-lane (the sums
// of the one-launch form), outputs written by that lane.  The queue holds at most one entry per sphere: it cannot
// overflow.
#ifndef MESH_HEAVY_FIRST
#define MESH_HEAVY_FIRST 1
#endif
#ifndef MESH_LONG_LIST
#define MESH_LONG_LIST 96  // entries in the list of the centre's cell from which a sphere counts as one of the launch's long chains
#endif
struct MeshQueueArgs {
  MeshCollArgs c;
  uint32_t *counter;  // workspace words 0 (entries from the head), 2 (entries from the tail), 1 (entries of queue2)
  uint2 *queue;       // workspace + 16 bytes: (sphere index, live slot mask)
  uint2 *queue2;      // behind it: the spheres the cell-list kernel hands to the tree walk
  int from_queue2;    // the walk kernel reads queue2 (filled from its head, counter word 1) instead of queue
};

#ifndef MESH_SELECT_STAGED
#define MESH_SELECT_STAGED 1
#endif
#ifndef MESH_SELECT_CHUNKS
#define MESH_SELECT_CHUNKS 4  // spheres per lane: 256 x 4 per workgroup (1 / 2 / 4 / 8 / 16: 47.6 / 44.5 / 39.6 / 48.4 / 68.2 us for 2.2 M spheres)
#endif
template <int SWEEP>
__global__ void __launch_bounds__(256) sphere_mesh_select_kernel(const MeshQueueArgs qa) {
  constexpr int CH = MESH_SELECT_CHUNKS;
  __shared__ int wave_cnt[2][CH * 4];  // live spheres of (class, chunk, wavefront); after the scan: their offset in the workgroup's run
  __shared__ uint32_t run_base[2];
  // The obstacle slots the workgroup's spheres meet (pose, root box of the mesh), fetched ONCE per workgroup into LDS (read per
  // sphere they are four dependent loads a slot: count / enable -> mesh id -> mesh record -> root box), and CH spheres a lane:
  // their loads are in flight together, and the two queue counters take one atomic per 256 x CH spheres.
  constexpr int kRecs = 128;
  __shared__ float s_rec[kRecs][16];  // t[3] q[4] lo[3] hi[3] enabled
  __shared__ MeshGridRec s_grid[kRecs];  // the slot's cell grid (mesh_cell_clear)
  const MeshCollArgs &a = qa.c;
  const long total = (long)a.batch * a.horizon * a.nspheres;  // (< 2^31: the launcher checks)
  const int tid = threadIdx.x, lane64 = tid & 63, wave = tid >> 6;
  const long base = (long)blockIdx.x * (256 * CH);
  const uint32_t hs = (uint32_t)(a.horizon * a.nspheres);
  const int first_b = (int)((uint32_t)base / hs);
  const int last_b = (int)((uint32_t)min(base + (256 * CH - 1), total - 1) / hs);
  const int n_rec = (a.use_multi_env ? last_b - first_b + 1 : 1) * a.nslots;
  const bool staged = MESH_SELECT_STAGED && n_rec <= kRecs;
  if (staged) {
    for (int r = tid; r < n_rec; r += 256) {
      const int bb = r / a.nslots, k = r - bb * a.nslots;
      const MeshSlot slot = load_mesh_slot(a.set, a.use_multi_env ? a.env_query_idx[first_b + bb] : 0, a.slot0 + k);
      float *rec = s_rec[r];
      rec[0] = slot.t.x; rec[1] = slot.t.y; rec[2] = slot.t.z;
      rec[3] = slot.qw; rec[4] = slot.qx; rec[5] = slot.qy; rec[6] = slot.qz;
      if (slot.enabled) {
        const float *rb = slot.m.node_box + 8;
        rec[7] = rb[0]; rec[8] = rb[1]; rec[9] = rb[2]; rec[10] = rb[4]; rec[11] = rb[5]; rec[12] = rb[6];
      }
      rec[13] = slot.enabled ? 1.0f : 0.0f;
      s_grid[r] = load_grid_rec(slot.m, slot.enabled);
    }
  }
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const float eta = a.eta[0];
  // ---- the lane's spheres and their neighbours, all loads first
  float4 s[CH];
  float reach[CH];  // half the longer step to a neighbour (+ rounding): how far a sweep sample can lie from the centre
  int bb[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const long sidx = base + c * 256 + tid;
    const bool in = sidx < total;
    const uint32_t s32 = in ? (uint32_t)sidx : 0u;
    bb[c] = (int)(s32 / hs);
    const int h = (int)((s32 - (uint32_t)bb[c] * hs) / (uint32_t)a.nspheres);
    s[c] = in ? sph[sidx] : make_float4(0.f, 0.f, 0.f, -1.0f);
    reach[c] = 2e-6f;
    if (SWEEP > 0) {
      const float4 ps = (in && h > 0) ? sph[sidx - a.nspheres] : s[c], ns = (in && h < a.horizon - 1) ? sph[sidx + a.nspheres] : s[c];
      const f3 center = make_f3(s[c].x, s[c].y, s[c].z);
      const f3 dp = make_f3(ps.x, ps.y, ps.z) - center, dn = make_f3(ns.x, ns.y, ns.z) - center;
      // (a missing neighbour stands in as the sphere itself: half step 0, as the walk kernel has it)
      reach[c] = fmaxf(0.5f * sqrtf(dot(dp, dp)), 0.5f * sqrtf(dot(dn, dn))) * 1.0001f + 2e-6f;
    }
  }
  if (staged) __syncthreads();
  uint32_t live[CH];
  unsigned heavy_bits = 0u;  // the centre lies inside the bounding box of a live slot's mesh: long walks (front of the queue)
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const long sidx = base + c * 256 + tid;
    const bool in = sidx < total;
    live[c] = 0u;
    bool heavy = false;
    if (in && s[c].w >= 0.0f) {
      const f3 center = make_f3(s[c].x, s[c].y, s[c].z);
      const float r_adj = s[c].w + eta;
      if (staged) {
        const float *recs = s_rec[a.use_multi_env ? (bb[c] - first_b) * a.nslots : 0];
        const float thr = r_adj + reach[c];
        for (int k = 0; k < a.nslots; k++) {
          const float *rec = recs + k * 16;
          if (rec[13] == 0.0f) continue;
          const f3 lc = quat_rot(rec[3], rec[4], rec[5], rec[6], center) + make_f3(rec[0], rec[1], rec[2]);  // (mesh_to_local)
          // (mesh_early_reject on the staged root box)
          const float ex = fmaxf(fmaxf(rec[7] - lc.x, lc.x - rec[10]), 0.0f), ey = fmaxf(fmaxf(rec[8] - lc.y, lc.y - rec[11]), 0.0f),
                      ez = fmaxf(fmaxf(rec[9] - lc.z, lc.z - rec[12]), 0.0f);
          int list_len;
          if (!(ex * ex + ey * ey + ez * ez > thr * thr * 1.00001f) &&
              !mesh_cell_clear(s_grid[(a.use_multi_env ? (bb[c] - first_b) * a.nslots : 0) + k], lc, thr, list_len)) {
            live[c] |= 1u << k;
            // the long chains of the launch: with cell lists the spheres whose centre's cell has a long list (inside a block every
            // face is about equally far), without them the spheres inside a mesh's bounding box
            if (MESH_HEAVY_FIRST)
              heavy = heavy || (list_len >= 0 ? list_len >= MESH_LONG_LIST
                                              : !(lc.x < rec[7] || lc.y < rec[8] || lc.z < rec[9] || lc.x > rec[10] || lc.y > rec[11] || lc.z > rec[12]));
          }
        }
      } else {
        const int env = a.use_multi_env ? a.env_query_idx[bb[c]] : 0;
        for (int k = 0; k < a.nslots; k++) {
          const MeshSlot slot = load_mesh_slot(a.set, env, a.slot0 + k);
          if (!slot.enabled) continue;
          const f3 lc = mesh_to_local(slot, center);
          int list_len;
          if (!mesh_early_reject(slot, lc, r_adj, reach[c]) && !mesh_cell_clear(load_grid_rec(slot.m, true), lc, r_adj + reach[c], list_len)) {
            live[c] |= 1u << k;
            const float *rb = slot.m.node_box + 8;
            if (MESH_HEAVY_FIRST)
              heavy = heavy || (list_len >= 0 ? list_len >= MESH_LONG_LIST
                                              : !(lc.x < rb[0] || lc.y < rb[1] || lc.z < rb[2] || lc.x > rb[4] || lc.y > rb[5] || lc.z > rb[6]));
          }
        }
      }
    }
    if (in && live[c] == 0u && !a.accumulate) {
#ifdef MESH_SELECT_PLAIN_STORES
      a.distance[sidx] = 0.0f;
      reinterpret_cast<float4 *>(a.gradient)[sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
#else  // written once, read by a later launch: past the caches' allocate-on-write
      __builtin_nontemporal_store(0.0f, a.distance + sidx);
      store_float4_streaming(reinterpret_cast<float4 *>(a.gradient) + sidx, make_float4(0.f, 0.f, 0.f, 0.f));
#endif
    }
    if (heavy) heavy_bits |= 1u << c;
  }
  // two classes: the heavy entries fill the queue from its head (counter word 0), the others from its tail backwards
  // (counter word 2): the walk takes the head first, so the launch's longest chains start when the launch does
  int before[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const bool is_h = live[c] != 0u && ((heavy_bits >> c) & 1u), is_l = live[c] != 0u && !((heavy_bits >> c) & 1u);
    const unsigned long long ball_h = __ballot(is_h), ball_l = __ballot(is_l);
    const unsigned long long ball = is_h ? ball_h : ball_l;
    before[c] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ball >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ball, 0u));
    if (lane64 == 0) { wave_cnt[0][c * 4 + wave] = __builtin_popcountll(ball_h); wave_cnt[1][c * 4 + wave] = __builtin_popcountll(ball_l); }
  }
  __syncthreads();
  if (tid < 2) {  // exclusive scan of the class's (chunk, wavefront) counts in place; one atomic for the workgroup's run
    int v[CH * 4], n = 0;
#pragma unroll
    for (int i = 0; i < CH * 4; i++) v[i] = wave_cnt[tid][i];
#pragma unroll
    for (int i = 0; i < CH * 4; i++) { wave_cnt[tid][i] = n; n += v[i]; }
    run_base[tid] = n ? atomicAdd(qa.counter + 2 * tid, (uint32_t)n) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CH; c++) {
    if (live[c] != 0u) {
      const bool is_h = ((heavy_bits >> c) & 1u) != 0u;
      const uint32_t pos = run_base[is_h ? 0 : 1] + (uint32_t)(wave_cnt[is_h ? 0 : 1][c * 4 + wave] + before[c]);
      qa.queue[is_h ? pos : (uint32_t)(total - 1) - pos] = make_uint2((uint32_t)(base + c * 256 + tid), live[c]);
    }
  }
}

// ---- the walk kernel: EIGHT LANES per live sphere (queue entry).  All eight run the sphere's scalar program (its slots in
// ascending order, every sample of a slot through the one query site of mesh_contribution) on the same values; only the
// tree walks split the work (mesh_device.hpp::mesh_contribution_group: three tree levels per step, one descendant box or
// one leaf triangle per lane, sibling keys kept in LDS).  Measured on the bench's mesh world (tools/r04/mesh_stats.py: 1024 x 33 points x 65
// spheres, the C2 world's 4 cuboids as meshes of 3072 triangles each, sweep 3), launch = select + walk:
//    one sphere per lane, binary walk            1430 us   (112 M VALU + 106 M SALU wave-instructions, ~150 of 8192
//                                                           wavefront slots busy on average: a few long walks per
//                                                           wavefront hold the other lanes)
//    eight lanes per sphere, leaves of 4 / of 8   801 / 677 us
//    + four wavefronts a SIMD (128 registers)      577 us
//    + sibling keys in LDS, nearest sibling next   460 us   (no box is fetched twice; half the dependent loads)
//    + one wavefront a workgroup                   420 us   (a wavefront's slot is free when ITS eight spheres are done,
//                                                           not when the slowest of 32 is; wavefront lifetimes: mean 67 us,
//                                                           the longest three times that)
//    sixteen lanes per sphere, leaves of 16        698 us   (before the last three steps; wider groups idle more lanes)
//    + every query of a sphere in ONE loop (a group settles a query and starts its next one while the others walk;
//      the lanes keep their own best triangle, the closest point is settled once per query)      402 us
// The walk kernel then makes 687 k passes through its loop (a wavefront: 43) for 2.64 M group moves and transitions: 3.85
// of a wavefront's eight groups are busy in a pass; ~310 instructions per pass, 4.1 SIMD cycles per instruction -- the
// rate every issue-bound kernel of this library runs at (2.9 - 4.4, profiles/r04_b_counters_by_kernel.json).
// Without effect: the closest-point-on-triangle test without branches (459 us against 460), five wavefronts a SIMD (421),
// a cap on the grid (the workgroups beyond the queue's end return at once).
// Work of that launch: 129 k of the 2.2 M spheres pass the select kernel, 332 k closest-point queries, 1.66 M interior steps,
// 0.57 M leaves (17 moves per sphere); the walk kernel is bound by instruction issue (182 M VALU + 102 M SALU wave-
// instructions: the eight spheres of a wavefront are each somewhere else in the program), not by the latency of the box
// fetches: more wavefronts per SIMD (96 / 80 registers, some spilled) change nothing, and neither does the while-while form
// (all eight move through interior nodes until each is at a leaf, then the leaves together: 531 us -- waiting costs what
// the shared leaf code saves).
// (Measured and dropped before that: the same work as ONE loop in which every lane runs its own program -- an explicit state
// machine around a walker shared by the closest-point and the ray mode, leaves one triangle per iteration, lanes refilled
// from the queue.  It keeps every lane busy, and executes the union of all states' code in every iteration: 334 M VALU +
// 233 M SALU wave-instructions per launch for 8 M node visits, 2.3 - 2.6 ms.)
#ifndef MESH_WALK_GROUP
#define MESH_WALK_GROUP 8
#endif
#ifndef MESH_WALK_THREADS
#define MESH_WALK_THREADS 64  // one wavefront a workgroup: its slot is free again when ITS eight spheres are done
#endif
#ifndef MESH_WALK_MAX_BLOCKS
#define MESH_WALK_MAX_BLOCKS (1 << 20)
#endif
#ifndef MESH_WALK_ATTR
#define MESH_WALK_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // 128 registers: four wavefronts a SIMD, nothing spilled
#endif
template <int SWEEP>
__global__ void __launch_bounds__(MESH_WALK_THREADS) MESH_WALK_ATTR sphere_mesh_walk_kernel(const MeshQueueArgs qa) {
  const MeshCollArgs &a = qa.c;
  // heavy entries from the head, the others from the tail; queue2 (the cell-list kernel's leftovers) from its head only
  const uint32_t n_front = qa.from_queue2 ? qa.counter[1] : qa.counter[0], n = qa.from_queue2 ? n_front : n_front + qa.counter[2];
  const uint2 *queue = qa.from_queue2 ? qa.queue2 : qa.queue;
  const uint32_t q_last = (uint32_t)((long)a.batch * a.horizon * a.nspheres - 1);
  const int hs = a.horizon * a.nspheres;
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const float w = a.weight[0], eta = a.eta[0];
  // lane 0 of a group writes
  constexpr unsigned G = MESH_WALK_GROUP, PER_WG = MESH_WALK_THREADS / G;
  __shared__ float group_keys[MESH_GROUP_LEVELS * MESH_WALK_THREADS];
  for (uint32_t q0 = blockIdx.x * PER_WG; q0 < n; q0 += gridDim.x * PER_WG) {
    const uint32_t q = q0 + threadIdx.x / G;
    // (a group beyond the end of the queue repeats the last entry so that ballots and shuffles stay whole; it writes nothing)
    const bool live_group = q < n;
    const uint32_t qq = live_group ? q : n - 1;
    const uint2 e = queue[qq < n_front ? qq : q_last - (qq - n_front)];
    if (live_group && (threadIdx.x & (G - 1u)) == 0) CUROBO_MESH_COUNT(6, 1);
    const long sidx = (long)e.x;
    const int b = (int)(e.x / (uint32_t)hs);  // (32-bit: the queued form holds sphere indices in 32 bits)
    const int h = (int)((e.x - (uint32_t)b * (uint32_t)hs) / (uint32_t)a.nspheres);
    const int env = a.use_multi_env ? a.env_query_idx[b] : 0;
    const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;
    const bool nb_prev = need_nb && h > 0, nb_next = need_nb && h < a.horizon - 1;
    const float4 s = sph[sidx];
    const float4 ps = nb_prev ? sph[sidx - a.nspheres] : s, ns = nb_next ? sph[sidx + a.nspheres] : s;
    const f3 center = make_f3(s.x, s.y, s.z), pp = make_f3(ps.x, ps.y, ps.z), np = make_f3(ns.x, ns.y, ns.z);
    const bool hp = SWEEP > 0 && nb_prev, hn = SWEEP > 0 && nb_next;
    const float r_adj = s.w + eta;
    float half_w_prev = 0.0f, half_w_next = 0.0f;
    if (SWEEP > 0) {
      if (hp) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (hn) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
    float dsum = 0.0f;
    f3 gsum = make_f3(0.f, 0.f, 0.f);
    uint32_t m = e.y;
#pragma unroll 1
    while (m) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      const MeshSlot slot = load_mesh_slot(a.set, env, a.slot0 + k);
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      if ((threadIdx.x & (G - 1u)) == 0) CUROBO_MESH_COUNT(7, 1);
      mesh_contribution_group<SWEEP, (int)G>(slot, a.set.gradient_mode, mesh_to_local(slot, center), hp, hn, pp, np, r_adj, eta, half_w_prev,
                                     half_w_next, reach, cost_sum, grad_local, group_keys + threadIdx.x, MESH_WALK_THREADS, q);
      if (cost_sum > 0.0f) {
        const f3 gw = mesh_to_world_vector(slot, grad_local);
        dsum += w * cost_sum;
        gsum = gsum + w * gw;
      }
    }
    if (a.enable_speed_metric && nb_prev && nb_next && dsum > 0.0f) mesh_speed_metric(center, pp, np, a.speed_dt[0], dsum, gsum);
    if (!live_group || (threadIdx.x & (G - 1u)) != 0) continue;
    float4 *grad = reinterpret_cast<float4 *>(a.gradient);
    if (a.accumulate) {
      if (dsum > 0.0f) {
        a.distance[sidx] += dsum;
        const float4 g0 = grad[sidx];
        grad[sidx] = make_float4(g0.x + gsum.x, g0.y + gsum.y, g0.z + gsum.z, g0.w);
      }
    } else {
      a.distance[sidx] = dsum;
      grad[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
    }
  }
}

// ---- the cell-list kernel: the walk kernel's job without a tree walk.  Eight lanes per live sphere again (a chunk of a
// cell's list is eight triangles, one per lane); every query of the sphere goes through mesh_cells_sdf.  A sphere with a query
// the lists cannot answer (outside the grid yet within reach, a cell without a list, a list that ends too early) is handed
// whole to the walk kernel through queue2 and writes nothing here.
#ifndef MESH_CELLS_GROUP
#define MESH_CELLS_GROUP 8  // lanes per live sphere (4 / 8 / 16 / 32 measured: 8)
#endif
#ifndef MESH_CELLS_UNROLL
#define MESH_CELLS_UNROLL 4  // list entries per lane and round (2 / 4 / 8 measured: 4)
#endif
#ifndef MESH_CELLS_HEAVY_GROUP
#define MESH_CELLS_HEAVY_GROUP 0  // > 0: the head of the queue by a launch of its own with this many lanes per sphere
#endif
#ifndef MESH_CELLS_HEAVY_UNROLL
#define MESH_CELLS_HEAVY_UNROLL 2
#endif
#ifndef MESH_CELLS_ATTR
#define MESH_CELLS_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // 128 registers (142 unconstrained: three wavefronts a SIMD): 179 -> 156 us
#endif
// PART 0: the whole queue; 1: the entries queued from its head (centre inside a live mesh's bounding box: the long lists of
// cells inside a surface); 2: the entries queued from its tail.
template <int SWEEP, int GROUP, int UNROLL, int PART>
__global__ void __launch_bounds__(256) MESH_CELLS_ATTR sphere_mesh_cells_kernel(const MeshQueueArgs qa) {
  const MeshCollArgs &a = qa.c;
  const uint32_t n_front = qa.counter[0], n_all = n_front + qa.counter[2];
  const uint32_t q_begin = PART == 2 ? n_front : 0u, n = PART == 1 ? n_front : n_all;
  const uint32_t q_last = (uint32_t)((long)a.batch * a.horizon * a.nspheres - 1);
  const int hs = a.horizon * a.nspheres;
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const float w = a.weight[0], eta = a.eta[0];
  constexpr unsigned G = GROUP, PER_WG = 256 / G;
  __shared__ float s_state[PER_WG][MESH_ST_WORDS];  // the groups' LDS records (mesh_device.hpp MESH_ST_*)
  float *st = s_state[threadIdx.x / G];
  for (uint32_t q0 = q_begin + blockIdx.x * PER_WG; q0 < n; q0 += gridDim.x * PER_WG) {
    const uint32_t q = q0 + threadIdx.x / G;
    const bool live_group = q < n;
    const uint32_t qq = live_group ? q : n - 1;
    const uint2 e = qa.queue[qq < n_front ? qq : q_last - (qq - n_front)];
    const long sidx = (long)e.x;
    const int b = (int)(e.x / (uint32_t)hs);
    const int h = (int)((e.x - (uint32_t)b * (uint32_t)hs) / (uint32_t)a.nspheres);
    const int env = a.use_multi_env ? a.env_query_idx[b] : 0;
    const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;
    const bool nb_prev = need_nb && h > 0, nb_next = need_nb && h < a.horizon - 1;
    float r_adj, reach;
    {
      const float4 s = sph[sidx];
      const float4 ps = nb_prev ? sph[sidx - a.nspheres] : s, ns = nb_next ? sph[sidx + a.nspheres] : s;
      const f3 center = make_f3(s.x, s.y, s.z), pp = make_f3(ps.x, ps.y, ps.z), np = make_f3(ns.x, ns.y, ns.z);
      r_adj = s.w + eta;
      float half_w_prev = 0.0f, half_w_next = 0.0f;
      if (SWEEP > 0) {
        if (nb_prev) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
        if (nb_next) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
      }
      reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
      st_store3(st, MESH_ST_CENTER, center); st_store3(st, MESH_ST_PREV, pp); st_store3(st, MESH_ST_NEXT, np);
      st[MESH_ST_HALF_PREV] = half_w_prev; st[MESH_ST_HALF_NEXT] = half_w_next;
      st[MESH_ST_DSUM] = 0.0f; st_store3(st, MESH_ST_GSUM, make_f3(0.f, 0.f, 0.f));
    }
    const unsigned flags = (SWEEP > 0 && nb_prev ? 1u : 0u) | (SWEEP > 0 && nb_next ? 2u : 0u);
    uint32_t m = e.y;
    int code = MESH_CELLS_OK;
#pragma unroll 1
    while (m && code == MESH_CELLS_OK) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      const MeshPoseSlot slot = load_mesh_pose_slot(a.set, env, a.slot0 + k, st);
      st[MESH_ST_COST] = 0.0f; st_store3(st, MESH_ST_GRAD, make_f3(0.f, 0.f, 0.f));
      code = mesh_contribution_cells<SWEEP, (int)G, UNROLL>(slot, a.set.gradient_mode, st, flags, r_adj, eta, reach, q);
      const float cost_sum = st[MESH_ST_COST];
      if (cost_sum > 0.0f) {
        const f3 gw = mesh_to_world_vector_st(st, st_load3(st, MESH_ST_GRAD));
        st[MESH_ST_DSUM] += w * cost_sum;
        st_store3(st, MESH_ST_GSUM, st_load3(st, MESH_ST_GSUM) + w * gw);
      }
    }
    if (!live_group || (threadIdx.x & (G - 1u)) != 0) continue;
    if (code != MESH_CELLS_OK) {  // whole, to the tree walk (queue2 from its head) or to a workgroup of its own (from its tail)
      if (code == MESH_CELLS_TO_WALK) qa.queue2[atomicAdd(qa.counter + 1, 1u)] = e;
      else qa.queue2[q_last - atomicAdd(qa.counter + 3, 1u)] = e;
      continue;
    }
    float dsum = st[MESH_ST_DSUM];
    f3 gsum = st_load3(st, MESH_ST_GSUM);
    if (a.enable_speed_metric && nb_prev && nb_next && dsum > 0.0f)
      mesh_speed_metric(st_load3(st, MESH_ST_CENTER), st_load3(st, MESH_ST_PREV), st_load3(st, MESH_ST_NEXT), a.speed_dt[0], dsum, gsum);
    float4 *grad = reinterpret_cast<float4 *>(a.gradient);
    if (a.accumulate) {
      if (dsum > 0.0f) {
        a.distance[sidx] += dsum;
        const float4 g0 = grad[sidx];
        grad[sidx] = make_float4(g0.x + gsum.x, g0.y + gsum.y, g0.z + gsum.z, g0.w);
      }
    } else {
      a.distance[sidx] = dsum;
      grad[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
    }
  }
}

// ---- one workgroup per sphere: what the cell-list kernel sent to the tail of queue2 (counter word 3) -- spheres with a query
// about equally far from hundreds or thousands of triangles (the middle of a ball, the axis of a pipe).  Eight lanes walking a
// tree that cannot prune, or scanning a list that is the whole mesh, held the launch for 430 us with FIVE such spheres in the
// tests' mesh world (128 trajectories: 473 us, of which the walk kernel 406; docs/NOTEBOOK.md round 6); 256 lanes answer each
// query by mesh_block_sdf.  Every thread of the workgroup carries the sphere's sweep state (the same values): the control flow
// is uniform, the barriers inside the query are met by all.
template <int SWEEP>
__global__ void __launch_bounds__(256) sphere_mesh_wide_kernel(const MeshQueueArgs qa) {
  const MeshCollArgs &a = qa.c;
  const uint32_t n = qa.counter[3];
  const uint32_t q_last = (uint32_t)((long)a.batch * a.horizon * a.nspheres - 1);
  const int hs = a.horizon * a.nspheres;
  const float4 *sph = reinterpret_cast<const float4 *>(a.spheres);
  const float w = a.weight[0], eta = a.eta[0];
  __shared__ MeshWideLds lds;
  for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
    const uint2 e = qa.queue2[q_last - q];
    const long sidx = (long)e.x;
    const int b = (int)(e.x / (uint32_t)hs);
    const int h = (int)((e.x - (uint32_t)b * (uint32_t)hs) / (uint32_t)a.nspheres);
    const int env = a.use_multi_env ? a.env_query_idx[b] : 0;
    const bool need_nb = SWEEP > 0 || a.enable_speed_metric != 0;
    const bool nb_prev = need_nb && h > 0, nb_next = need_nb && h < a.horizon - 1;
    const float4 s = sph[sidx];
    const float4 ps = nb_prev ? sph[sidx - a.nspheres] : s, ns = nb_next ? sph[sidx + a.nspheres] : s;
    const f3 center = make_f3(s.x, s.y, s.z), pp = make_f3(ps.x, ps.y, ps.z), np = make_f3(ns.x, ns.y, ns.z);
    const bool hp = SWEEP > 0 && nb_prev, hn = SWEEP > 0 && nb_next;
    const float r_adj = s.w + eta;
    float half_w_prev = 0.0f, half_w_next = 0.0f;
    if (SWEEP > 0) {
      if (hp) { const f3 dd = pp - center; half_w_prev = 0.5f * sqrtf(dot(dd, dd)); }
      if (hn) { const f3 dd = np - center; half_w_next = 0.5f * sqrtf(dot(dd, dd)); }
    }
    const float reach = SWEEP > 0 ? fmaxf(half_w_prev, half_w_next) * 1.0001f + 2e-6f : 2e-6f;
    float dsum = 0.0f;
    f3 gsum = make_f3(0.f, 0.f, 0.f);
    uint32_t m = e.y;
#pragma unroll 1
    while (m) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      const MeshSlot slot = load_mesh_slot(a.set, env, a.slot0 + k);
      float cost_sum = 0.0f;
      f3 grad_local = make_f3(0.f, 0.f, 0.f);
      mesh_contribution_q<SWEEP>(slot, a.set.gradient_mode, mesh_to_local(slot, center), hp, hn, pp, np, r_adj, eta, half_w_prev, half_w_next,
                                 reach, cost_sum, grad_local, [&](f3 qp, float, float max_distance, bool, f3 &g) {
                                   return mesh_block_sdf(slot.m, qp, max_distance, g, lds);
                                 });
      if (cost_sum > 0.0f) {
        const f3 gw = mesh_to_world_vector(slot, grad_local);
        dsum += w * cost_sum;
        gsum = gsum + w * gw;
      }
    }
    if (a.enable_speed_metric && nb_prev && nb_next && dsum > 0.0f) mesh_speed_metric(center, pp, np, a.speed_dt[0], dsum, gsum);
    if (threadIdx.x != 0) continue;
    float4 *grad = reinterpret_cast<float4 *>(a.gradient);
    if (a.accumulate) {
      if (dsum > 0.0f) {
        a.distance[sidx] += dsum;
        const float4 g0 = grad[sidx];
        grad[sidx] = make_float4(g0.x + gsum.x, g0.y + gsum.y, g0.z + gsum.z, g0.w);
      }
    } else {
      a.distance[sidx] = dsum;
      grad[sidx] = make_float4(gsum.x, gsum.y, gsum.z, 0.0f);
    }
  }
}

// the queue counters of a launch (workspace words 0-3) back to zero.  A kernel, not hipMemsetAsync: a captured memset node of
// these 16 bytes replays correctly ONCE -- the second replay of the graph faults (tools/r06/mesh_graph_replay.py,
// docs/NOTEBOOK.md round 6) -- and a solver replays its graphs for as long as it lives.
__global__ void __launch_bounds__(64) mesh_queue_reset_kernel(uint32_t *counter) {
  if (threadIdx.x < 4) counter[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) mesh_zero_outputs_kernel(float *distance, float4 *gradient, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) { distance[i] = 0.0f; gradient[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
}

}  // namespace curobo_hip

using namespace curobo_hip;

#ifdef CUROBO_MESH_STATS
CUROBO_EXPORT int curobo_hip_mesh_stats(unsigned long long *out_host8, int reset) {
  if (out_host8 && hipMemcpyFromSymbol(out_host8, HIP_SYMBOL(g_mesh_stats), 64) != hipSuccess) return CUROBO_HIP_ERR_LAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mesh_stats), z, 64) != hipSuccess) return CUROBO_HIP_ERR_LAUNCH; }
  return CUROBO_HIP_OK;
}
CUROBO_EXPORT int curobo_hip_mesh_lane_stats(unsigned int *out_host, int reset) {  // [1 << 18]
  if (out_host && hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_mesh_lane), sizeof(unsigned int) << 18) != hipSuccess) return CUROBO_HIP_ERR_LAUNCH;
  if (reset) { void *p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_mesh_lane)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned int) << 18) != hipSuccess) return CUROBO_HIP_ERR_LAUNCH; }
  return CUROBO_HIP_OK;
}
#endif

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
  // the stackless walks keep one bit per tree level in two 32-bit trails (mesh_device.hpp): depth <= 24 here
  CUROBO_REQUIRE(m->n_leaves <= (1 << 24), "%s: mesh too large", what);
  return CUROBO_HIP_OK;
}

static int check_grid(const curobo_hip_mesh *m, const char *what) {
  CUROBO_REQUIRE(m->grid_h > 0.0f && m->grid_n[0] > 0 && m->grid_n[1] > 0 && m->grid_n[2] > 0 && m->grid_pad >= 0.0f &&
                     (long)m->grid_n[0] * m->grid_n[1] * m->grid_n[2] < (1l << 30),
                 "%s: the mesh's grid fields (grid_lo / grid_h / grid_n / grid_pad) are not set", what);
  return CUROBO_HIP_OK;
}

CUROBO_EXPORT int curobo_hip_mesh_cells_count(int32_t *out_count, float *out_cover, uint8_t *out_side, float *out_centre_dist,
                                              const curobo_hip_mesh *mesh, int gather_cap, curobo_hip_stream_t stream) {
  const char *what = "mesh_cells_count";
  CUROBO_REQUIRE(out_count && out_cover && out_side && out_centre_dist && gather_cap >= 1, "%s: bad arguments", what);
  if (int rc = check_mesh(mesh, what)) return rc;
  if (int rc = check_grid(mesh, what)) return rc;
  MeshCellsArgs a{};
  a.m = *mesh; a.count = out_count; a.cover = out_cover; a.side = out_side; a.centre_dist = out_centre_dist; a.gather_cap = gather_cap;
  a.n_cells = mesh->grid_n[0] * mesh->grid_n[1] * mesh->grid_n[2];
  hipLaunchKernelGGL(mesh_cells_count_kernel, dim3((unsigned)ceil_div(a.n_cells, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch(what, (hipStream_t)stream);
}

CUROBO_EXPORT int curobo_hip_mesh_cells_fill(int64_t *out_keys, int32_t *out_entries, uint32_t *out_cell_start, const int64_t *offsets,
                                             const float *cover, const uint8_t *side, const float *centre_dist,
                                             const curobo_hip_mesh *mesh, curobo_hip_stream_t stream) {
  const char *what = "mesh_cells_fill";
  CUROBO_REQUIRE(out_keys && out_entries && out_cell_start && offsets && cover && side && centre_dist, "%s: NULL argument", what);
  if (int rc = check_mesh(mesh, what)) return rc;
  if (int rc = check_grid(mesh, what)) return rc;
  MeshCellsArgs a{};
  a.m = *mesh; a.cover = const_cast<float *>(cover); a.offsets = offsets; a.keys = out_keys; a.entries = out_entries;
  a.n_cells = mesh->grid_n[0] * mesh->grid_n[1] * mesh->grid_n[2];
  hipLaunchKernelGGL(mesh_cells_fill_kernel, dim3((unsigned)ceil_div(a.n_cells, 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(mesh_cells_start_kernel, dim3((unsigned)ceil_div(a.n_cells + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint2 *>(out_cell_start), offsets, side, centre_dist, a.n_cells);
  return check_launch(what, (hipStream_t)stream);
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

static int sphere_mesh_collision_impl(
    const char *what, float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres, int use_multi_env,
    int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate, void *workspace, size_t workspace_bytes,
    curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(distance && gradient && spheres && meshes && weight && activation_distance, "%s: NULL argument", what);
  CUROBO_REQUIRE(sweep_steps == 0 || sweep_steps == 3, "%s: sweep_steps must be 0 or 3 (reference SWEEP_STEPS)", what);
  CUROBO_REQUIRE(!use_multi_env || env_query_idx, "%s: use_multi_env needs env_query_idx", what);
  CUROBO_REQUIRE(!enable_speed_metric || speed_dt, "%s: speed metric needs speed_dt", what);
  CUROBO_REQUIRE(meshes->max_n == 0 || (meshes->meshes && meshes->mesh_id && meshes->inv_pose && meshes->dims && meshes->enable && meshes->count),
                 "%s: incomplete mesh set", what);
  const long total = (long)batch_size * horizon * num_spheres;
  if (total == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)ceil_div_l(total, 256)), block(256), select_grid((unsigned)ceil_div_l(total, 256 * MESH_SELECT_CHUNKS));
  if (meshes->max_n == 0) {  // no mesh slots: this kind's share is zero -- which an overwriting launch still has to write
    if (!accumulate) hipLaunchKernelGGL(mesh_zero_outputs_kernel, grid, block, 0, st, distance, reinterpret_cast<float4 *>(gradient), total);
    return check_launch(what, st);
  }
  MeshCollArgs a{};
  a.distance = distance; a.gradient = gradient; a.spheres = spheres; a.set = *meshes; a.weight = weight; a.eta = activation_distance;
  a.speed_dt = speed_dt; a.env_query_idx = env_query_idx; a.batch = batch_size; a.horizon = horizon; a.nspheres = num_spheres;
  a.use_multi_env = use_multi_env; a.enable_speed_metric = enable_speed_metric;
  const bool queued = workspace != nullptr;
  if (queued) {
    CUROBO_REQUIRE(total < (1l << 31), "%s: the queued form indexes spheres with 32 bits", what);
    CUROBO_REQUIRE(((uintptr_t)workspace & 15) == 0 && workspace_bytes >= 16 + 2 * (size_t)total * sizeof(uint2),
                   "%s: workspace too small or misaligned (curobo_hip_sphere_mesh_collision_ws_bytes)", what);
  }
  const bool with_cells = (meshes->flags & CUROBO_HIP_MESH_SET_HAS_CELLS) != 0;
  // 32 obstacle slots per launch (the live mask of a sphere); further groups add to the first one's output
  for (int slot0 = 0; slot0 < meshes->max_n; slot0 += kMeshSlotsPerLaunch) {
    a.slot0 = slot0;
    a.nslots = std::min(kMeshSlotsPerLaunch, meshes->max_n - slot0);
    a.accumulate = (accumulate || slot0 > 0) ? 1 : 0;
    if (queued) {
      MeshQueueArgs qa{};
      qa.c = a; qa.counter = reinterpret_cast<uint32_t *>(workspace);
      qa.queue = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(workspace) + 16);
      qa.queue2 = qa.queue + total;
      hipLaunchKernelGGL(mesh_queue_reset_kernel, dim3(1), dim3(64), 0, st, qa.counter);
      // the walk's grid covers the chip once (1024 workgroups of four wavefronts); the queue is usually much shorter
      const unsigned walk_blocks = (unsigned)std::min<long>(MESH_WALK_MAX_BLOCKS, ceil_div_l(total, MESH_WALK_THREADS / MESH_WALK_GROUP));
      // with cell lists: select -> cell-list kernel (every sphere it can answer) -> tree walk of the few it could not
      const unsigned cells_blocks = (unsigned)std::min<long>(8192, ceil_div_l(total, 256 / MESH_CELLS_GROUP));
      const unsigned rest_blocks = std::min(walk_blocks, 4096u);
      const unsigned wide_blocks = (unsigned)std::min<long>(2048, total);  // (a workgroup per sphere sent there: usually none)
      if (sweep_steps > 0) {
        hipLaunchKernelGGL((sphere_mesh_select_kernel<3>), select_grid, block, 0, st, qa);
        if (with_cells) {
          if (MESH_CELLS_HEAVY_GROUP > 0) {
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<3, MESH_CELLS_HEAVY_GROUP ? MESH_CELLS_HEAVY_GROUP : 8, MESH_CELLS_HEAVY_UNROLL, 1>), dim3(cells_blocks), block, 0, st, qa);
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<3, MESH_CELLS_GROUP, MESH_CELLS_UNROLL, 2>), dim3(cells_blocks), block, 0, st, qa);
          } else {
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<3, MESH_CELLS_GROUP, MESH_CELLS_UNROLL, 0>), dim3(cells_blocks), block, 0, st, qa);
          }
          qa.from_queue2 = 1;
          hipLaunchKernelGGL((sphere_mesh_wide_kernel<3>), dim3(wide_blocks), dim3(256), 0, st, qa);
          hipLaunchKernelGGL((sphere_mesh_walk_kernel<3>), dim3(rest_blocks), dim3(MESH_WALK_THREADS), 0, st, qa);
        } else {
          hipLaunchKernelGGL((sphere_mesh_walk_kernel<3>), dim3(walk_blocks), dim3(MESH_WALK_THREADS), 0, st, qa);
        }
      } else {
        hipLaunchKernelGGL((sphere_mesh_select_kernel<0>), select_grid, block, 0, st, qa);
        if (with_cells) {
          if (MESH_CELLS_HEAVY_GROUP > 0) {
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<0, MESH_CELLS_HEAVY_GROUP ? MESH_CELLS_HEAVY_GROUP : 8, MESH_CELLS_HEAVY_UNROLL, 1>), dim3(cells_blocks), block, 0, st, qa);
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<0, MESH_CELLS_GROUP, MESH_CELLS_UNROLL, 2>), dim3(cells_blocks), block, 0, st, qa);
          } else {
            hipLaunchKernelGGL((sphere_mesh_cells_kernel<0, MESH_CELLS_GROUP, MESH_CELLS_UNROLL, 0>), dim3(cells_blocks), block, 0, st, qa);
          }
          qa.from_queue2 = 1;
          hipLaunchKernelGGL((sphere_mesh_wide_kernel<0>), dim3(wide_blocks), dim3(256), 0, st, qa);
          hipLaunchKernelGGL((sphere_mesh_walk_kernel<0>), dim3(rest_blocks), dim3(MESH_WALK_THREADS), 0, st, qa);
        } else {
          hipLaunchKernelGGL((sphere_mesh_walk_kernel<0>), dim3(walk_blocks), dim3(MESH_WALK_THREADS), 0, st, qa);
        }
      }
      continue;
    }
    const size_t lds = (size_t)(3 + 1) * 256 * 16 + (256 + 8) * 4 + (((size_t)256 * a.nslots * 2 + 15) & ~(size_t)15);
    if (sweep_steps > 0) hipLaunchKernelGGL((sphere_mesh_collision_kernel<3>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((sphere_mesh_collision_kernel<0>), grid, block, lds, st, a);
  }
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_sphere_mesh_collision(
    float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres, int use_multi_env,
    int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate, curobo_hip_stream_t stream) {
  return sphere_mesh_collision_impl("sphere_mesh_collision", distance, gradient, spheres, meshes, weight, activation_distance, env_query_idx,
                                    batch_size, horizon, num_spheres, use_multi_env, sweep_steps, enable_speed_metric, speed_dt, accumulate,
                                    nullptr, 0, stream);
}

CUROBO_EXPORT int curobo_hip_sphere_mesh_collision_ws_bytes(int batch_size, int horizon, int num_spheres, int64_t *out_bytes_host) {
  CUROBO_REQUIRE(out_bytes_host && batch_size >= 0 && horizon >= 0 && num_spheres >= 0, "sphere_mesh_collision_ws_bytes: bad arguments%s", "");
  *out_bytes_host = 16 + 2 * (int64_t)batch_size * horizon * num_spheres * (int64_t)sizeof(uint2);  // counters, queue, queue2
  return CUROBO_HIP_OK;
}

CUROBO_EXPORT int curobo_hip_sphere_mesh_collision_ws(
    float *distance, float *gradient, const float *spheres, const curobo_hip_mesh_set *meshes, const float *weight,
    const float *activation_distance, const int32_t *env_query_idx, int batch_size, int horizon, int num_spheres, int use_multi_env,
    int sweep_steps, int enable_speed_metric, const float *speed_dt, int accumulate, void *workspace, size_t workspace_bytes,
    curobo_hip_stream_t stream) {
  const char *what = "sphere_mesh_collision_ws";
  CUROBO_REQUIRE(workspace, "%s: NULL workspace", what);
  return sphere_mesh_collision_impl(what, distance, gradient, spheres, meshes, weight, activation_distance, env_query_idx, batch_size, horizon,
                                    num_spheres, use_multi_env, sweep_steps, enable_speed_metric, speed_dt, accumulate, workspace,
                                    workspace_bytes, stream);
}
