// mesh_bake.hip -- triangle mesh -> fp16 ESDF voxel grid, on the device, at scene-upload time.
//
// The reference routes mesh obstacles (and every non-cuboid primitive) through NVIDIA Warp's BVH
// (geom/data/data_mesh.py:555-700: wp.mesh_query_point per query sphere, sign from the face normal / winding);
// Warp has no ROCm backend.  Here a mesh becomes one more ESDF grid of the voxel store the collision kernels
// already read (layout of the reference's VoxelGrid / VoxelData, geom/data/data_voxel.py:42-95): one lane per voxel,
// triangles streamed through LDS in tiles, exact point-triangle distance (Ericson, Real-Time Collision Detection
// 5.1.5) and the sign from the generalised winding number (sum of van Oosterom-Strackee solid angles; robust at
// edges / vertices for closed, consistently oriented meshes).  Same algorithm as the NumPy reference
// curobo_amd/scene/primitives.py::mesh_sdf (the checker of tests/test_gpu_kernels.py::test_mesh_esdf_bake_on_device).
// O(voxels x triangles): 128^3 voxels x 1 k triangles = 2e9 tests, a few milliseconds once per world update.
#include "common.hpp"

#include <hip/hip_fp16.h>

namespace curobo_hip {

struct MeshBakeArgs {
  __half *out;           // [nx * ny * nz]
  const float *vertices; // [V][3] mesh frame
  const int32_t *faces;  // [F][3]
  int n_faces, nx, ny, nz;
  float voxel_size, max_distance;
  float g2m[12];         // grid frame -> mesh frame, row-major 3x4
};

constexpr int kBakeTile = 128;

__device__ __forceinline__ float point_triangle_dist2(f3 p, f3 a, f3 ab, f3 ac) {
  const f3 ap = p - a;
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  f3 closest;
  if (d1 <= 0.0f && d2 <= 0.0f) closest = a;
  else {
    const f3 b = a + ab, bp = p - b;
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) closest = b;
    else {
      const float vc = d1 * d4 - d3 * d2;
      if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) closest = a + (d1 / (d1 - d3)) * ab;
      else {
        const f3 c = a + ac, cp = p - c;
        const float d5 = dot(ab, cp), d6 = dot(ac, cp);
        if (d6 >= 0.0f && d5 <= d6) closest = c;
        else {
          const float vb = d5 * d2 - d1 * d6;
          if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) closest = a + (d2 / (d2 - d6)) * ac;
          else {
            const float va = d3 * d6 - d5 * d4;
            if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) closest = b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
            else {
              const float den = 1.0f / (va + vb + vc);
              closest = a + (vb * den) * ab + (vc * den) * ac;
            }
          }
        }
      }
    }
  }
  const f3 d = p - closest;
  return dot(d, d);
}

__global__ void __launch_bounds__(256) mesh_esdf_bake_kernel(const MeshBakeArgs a) {
  __shared__ float tri[kBakeTile][9];  // a, b - a, c - a
  const long n_vox = (long)a.nx * a.ny * a.nz;
  const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = v < n_vox;
  f3 p = make_f3(0.f, 0.f, 0.f);
  if (in) {
    const int iz = (int)(v % a.nz), iy = (int)((v / a.nz) % a.ny), ix = (int)(v / ((long)a.nz * a.ny));
    // voxel centres as the collision kernels read them back: (i + 0.5 - n / 2) * voxel_size in the grid frame
    const f3 g = make_f3(((float)ix + 0.5f - 0.5f * (float)a.nx) * a.voxel_size, ((float)iy + 0.5f - 0.5f * (float)a.ny) * a.voxel_size,
                         ((float)iz + 0.5f - 0.5f * (float)a.nz) * a.voxel_size);
    p = make_f3(a.g2m[0] * g.x + a.g2m[1] * g.y + a.g2m[2] * g.z + a.g2m[3], a.g2m[4] * g.x + a.g2m[5] * g.y + a.g2m[6] * g.z + a.g2m[7],
                a.g2m[8] * g.x + a.g2m[9] * g.y + a.g2m[10] * g.z + a.g2m[11]);
  }
  float best = 3.0e38f, solid = 0.0f;
  for (int t0 = 0; t0 < a.n_faces; t0 += kBakeTile) {
    const int cnt = min(kBakeTile, a.n_faces - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 9; e += blockDim.x) {
      const int f = e / 9, k = e - f * 9, vert = k / 3, ax = k - vert * 3;
      const float va = a.vertices[(size_t)a.faces[(size_t)(t0 + f) * 3] * 3 + ax];
      const float vv = a.vertices[(size_t)a.faces[(size_t)(t0 + f) * 3 + vert] * 3 + ax];
      tri[f][k] = vert == 0 ? va : vv - va;
    }
    __syncthreads();
    if (in)
      for (int f = 0; f < cnt; f++) {
        const f3 A = make_f3(tri[f][0], tri[f][1], tri[f][2]), ab = make_f3(tri[f][3], tri[f][4], tri[f][5]),
                 ac = make_f3(tri[f][6], tri[f][7], tri[f][8]);
        best = fminf(best, point_triangle_dist2(p, A, ab, ac));
        const f3 ra = A - p, rb = ra + ab, rc = ra + ac;
        const float la = sqrtf(dot(ra, ra)), lb = sqrtf(dot(rb, rb)), lc = sqrtf(dot(rc, rc));
        const float num = dot(ra, cross(rb, rc));
        const float den = la * lb * lc + dot(ra, rb) * lc + dot(rb, rc) * la + dot(rc, ra) * lb;
        solid += 2.0f * atan2f(num, den);
      }
  }
  if (!in) return;
  const float d = sqrtf(best);
  const bool inside = fabsf(solid) > 6.2831853f;  // |winding number| > 0.5
  float sdf = inside ? -d : d;
  sdf = fminf(fmaxf(sdf, -a.max_distance), a.max_distance);
  a.out[v] = __float2half(sdf);
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_mesh_esdf_bake(uint16_t *out_esdf_fp16, const float *vertices, const int32_t *faces, int n_vertices,
                                            int n_faces, int nx, int ny, int nz, float voxel_size, float max_distance,
                                            const float *grid_to_mesh_3x4_host, curobo_hip_stream_t stream) {
  const char *what = "mesh_esdf_bake";
  CUROBO_REQUIRE(out_esdf_fp16 && vertices && faces && grid_to_mesh_3x4_host, "%s: NULL argument", what);
  CUROBO_REQUIRE(n_vertices > 0 && n_faces > 0 && nx > 0 && ny > 0 && nz > 0 && voxel_size > 0.0f, "%s: empty mesh or grid", what);
  MeshBakeArgs a{};
  a.out = reinterpret_cast<__half *>(out_esdf_fp16); a.vertices = vertices; a.faces = faces; a.n_faces = n_faces;
  a.nx = nx; a.ny = ny; a.nz = nz; a.voxel_size = voxel_size; a.max_distance = max_distance;
  for (int i = 0; i < 12; i++) a.g2m[i] = grid_to_mesh_3x4_host[i];
  const long n_vox = (long)nx * ny * nz;
  hipLaunchKernelGGL(mesh_esdf_bake_kernel, dim3((unsigned)ceil_div_l(n_vox, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch(what, (hipStream_t)stream);
}
