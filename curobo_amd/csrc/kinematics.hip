// kinematics.hip -- forward kinematics over the URDF tree (+ collision spheres, geometric
// Jacobian, centre of mass) and its VJP, for gfx950 (wave64).
//
// What is computed follows the reference kernels
//   curobolib/kernels/kinematics/kinematics_forward_kernel.cuh:20-433
//   curobolib/kernels/kinematics/kinematics_forward_helper.cuh:45-600
//   curobolib/kernels/kinematics/kinematics_backward_kernel.cuh:27-157
//   curobolib/kernels/kinematics/kinematics_backward_helper.cuh:14-291
// How it is computed is CDNA4-specific:
//   * a point (one joint configuration) is owned by 16 consecutive lanes, 4 points per wave64;
//     lane (4*r + c) of the group owns element (r, c) of the 3x4 link transform.  The serial
//     chain  C_l = C_parent(l) * M_l  needs, per lane, the three rotation entries of ITS OWN row
//     of the parent: they live in the same DPP quad, so each chain step is 4 quad broadcasts +
//     one 16-byte LDS read of the local-transform column + 4 FMAs.  A lane only ever re-reads
//     cumulative entries that it wrote itself, so the whole chain runs without any barrier
//     (the reference needs a __syncwarp per link).
//   * local joint transforms (the sincos) are computed once per (point, link) by all 256 lanes,
//     column-major and padded to float4 so the chain reads them with ds_read_b128.
//   * cumulative transforms are built directly in the [L][3][4] output layout in LDS and leave
//     the CU as one contiguous float4 stream per block; spheres leave as 16 lanes x 16 B = 256 B
//     contiguous segments per point.
//   * the VJP accumulates per-lane partial joint gradients in LDS rows (ds_add_f32, no dynamic
//     register indexing -> no scratch), then 16-lane reduces them.
#include <cstdlib>

#include "fk_device.hpp"

namespace curobo_hip {

struct FkArgs {
  float *link_pos;
  float *link_quat;
  float *spheres_out;
  float *com_out;
  float *jacobian_out;
  float *cumul_out;
  const float *q;
  const float *fixed_transform;
  const float *robot_spheres;
  const float *link_masses_com;
  const int8_t *joint_map_type;
  const int16_t *joint_map;
  const int16_t *link_map;
  const int16_t *tool_frame_map;
  const int16_t *link_sphere_map;
  const int16_t *link_chain_data;
  const int16_t *link_chain_offsets;
  const int16_t *joint_links_data;
  const int16_t *joint_links_offsets;
  const uint8_t *joint_affects_endeffector;
  const float *joint_offset;
  const int32_t *env_query_idx;
  int n_points, horizon, nspheres, num_envs, nlinks, njoints, n_tool_frames;
};

constexpr int kFkLinkStride = 16;  // floats of LDS per (point, link): the local 3x4 column-major padded, then the
                                   // cumulative 3x4 row-major over it (fk_chain_16_inplace)

template <bool SPHERES, bool JACOBIAN, bool COM, bool WRITE_CUMUL>
__global__ void __launch_bounds__(256) fk_forward_kernel(const FkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.nlinks;
  const int pts = blockDim.x / kFkLanes;
  float *cumul = smem;                                                    // [pts][L][16]
  float4 *s_sph = reinterpret_cast<float4 *>(smem + pts * L * kFkLinkStride);  // [S] robot spheres (one env)
  int *s_sph_link = reinterpret_cast<int *>(s_sph + (SPHERES ? a.nspheres : 0));  // [S]
  int *s_parent = s_sph_link + (SPHERES ? a.nspheres : 0);                // [L]

  const int tid = threadIdx.x;
  const int pt0 = blockIdx.x * pts;
  const int npts = min(pts, a.n_points - pt0);

  for (int l = tid; l < L; l += blockDim.x) s_parent[l] = a.link_map[l];
  // one set of robot spheres: the table goes to LDS with the first loads of the block, and the sphere pass
  // below is LDS -> registers -> HBM with no global load on its path
  const bool spheres_staged = SPHERES && a.num_envs <= 1;
  if (spheres_staged) {
    for (int s = tid; s < a.nspheres; s += blockDim.x) {
      s_sph[s] = reinterpret_cast<const float4 *>(a.robot_spheres)[s];
      s_sph_link[s] = a.link_sphere_map[s];
    }
  }

  // ---- phase 1: local transforms, one (point, link) item per lane
  for (int e = tid; e < npts * L; e += blockDim.x) {
    const int lp = e / L;
    const int l = e - lp * L;
    const int jt = a.joint_map_type[l];
    float qv = 0.0f;
    if (jt != J_FIXED) qv = a.q[(size_t)(pt0 + lp) * a.njoints + a.joint_map[l]];
    local_transform_colmajor(cumul + (size_t)e * kFkLinkStride, a.fixed_transform + l * 12, jt, qv,
                             a.joint_offset[2 * l], a.joint_offset[2 * l + 1]);
  }
  __syncthreads();

  // ---- phase 2: serial chain, barrier free (see file header)
  const int grp = tid / kFkLanes;
  const int lane = tid % kFkLanes;
  const bool active = grp < npts;
  float *my_cumul = cumul + (size_t)grp * L * kFkLinkStride;
  if (active) fk_chain_16_inplace(my_cumul, s_parent, a.fixed_transform, L, lane);
  __syncthreads();

  // ---- phase 3a: cumulative transforms -> HBM as [L][3][4], one contiguous float4 stream per block
  if (WRITE_CUMUL) {
    const float4 *src = reinterpret_cast<const float4 *>(cumul);
    float4 *dst = reinterpret_cast<float4 *>(a.cumul_out + (size_t)pt0 * L * 12);
    for (int i = tid; i < npts * L * 3; i += blockDim.x) {
      const int link = i / 3;
      store_float4_streaming(dst + i, src[link * 4 + (i - link * 3)]);
    }
  }
  if (!active) return;
  const int n = pt0 + grp;

  // ---- phase 3b: collision spheres (reference kinematics_forward_helper.cuh:218-254)
  if (SPHERES) {
    float4 *out = reinterpret_cast<float4 *>(a.spheres_out) + (size_t)n * a.nspheres;
    if (spheres_staged) {
      for (int s = lane; s < a.nspheres; s += kFkLanes)
        store_float4_streaming(out + s, transform_sphere(my_cumul + s_sph_link[s] * kFkLinkStride, s_sph[s]));
    } else {
      const int env = a.env_query_idx[n / a.horizon];
      const float4 *rs = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)env * a.nspheres;
      for (int s = lane; s < a.nspheres; s += kFkLanes)
        out[s] = transform_sphere(my_cumul + a.link_sphere_map[s] * kFkLinkStride, rs[s]);
    }
  }
  // ---- centre of mass (reference :538-600)
  if (COM) {
    float wx = 0.f, wy = 0.f, wz = 0.f, wm = 0.f;
    for (int l = lane; l < L; l += kFkLanes) {
      const float4 mc = reinterpret_cast<const float4 *>(a.link_masses_com)[l];
      if (mc.w > 0.0f) {
        const float4 cw = transform_sphere(my_cumul + l * kFkLinkStride, mc);
        wx += mc.w * cw.x; wy += mc.w * cw.y; wz += mc.w * cw.z; wm += mc.w;
      }
    }
    wx = group_sum<kFkLanes>(wx); wy = group_sum<kFkLanes>(wy);
    wz = group_sum<kFkLanes>(wz); wm = group_sum<kFkLanes>(wm);
    if (lane == 0) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wm > 0.0f) o = make_float4(wx / wm, wy / wm, wz / wm, wm);
      reinterpret_cast<float4 *>(a.com_out)[n] = o;
    }
  }
  // ---- tool-frame poses (reference :270-301), quaternion written wxyz
  for (int t = lane; t < a.n_tool_frames; t += kFkLanes) {
    const float *C = my_cumul + a.tool_frame_map[t] * kFkLinkStride;
    const float4 qx = quat_from_transform(C);
    reinterpret_cast<float4 *>(a.link_quat)[(size_t)n * a.n_tool_frames + t] = make_float4(qx.w, qx.x, qx.y, qx.z);
    float *p = a.link_pos + ((size_t)n * a.n_tool_frames + t) * 3;
    p[0] = C[3]; p[1] = C[7]; p[2] = C[11];
  }
  // ---- geometric Jacobian (reference :45-200), one joint column per lane, no atomics
  if (JACOBIAN) {
    const int D = a.njoints, T = a.n_tool_frames;
    for (int t = 0; t < T; t++) {
      const int tl = a.tool_frame_map[t];
      const float *E = my_cumul + tl * kFkLinkStride;
      const f3 ee = make_f3(E[3], E[7], E[11]);
      const int cs = a.link_chain_offsets[tl], ce = a.link_chain_offsets[tl + 1];
      float *J = a.jacobian_out + ((size_t)n * T + t) * 6 * D;
      for (int j = lane; j < D; j += kFkLanes) {
        float col[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.joint_affects_endeffector[j * T + t]) {
          for (int jl = a.joint_links_offsets[j]; jl < a.joint_links_offsets[j + 1]; jl++) {
            const int li = a.joint_links_data[jl];
            if (li == 0) continue;
            bool in_chain = false;
            for (int ci = cs; ci < ce; ci++) in_chain |= (a.link_chain_data[ci] == li);
            if (!in_chain) continue;
            const float *C = my_cumul + li * kFkLinkStride;
            const int jt = a.joint_map_type[li];
            const float sign = a.joint_offset[li * 2];
            if (jt >= J_X_ROT) {
              const int ax = jt - J_X_ROT;
              const f3 axis = sign * make_f3(C[ax], C[4 + ax], C[8 + ax]);
              const f3 lin = cross(axis, ee - make_f3(C[3], C[7], C[11]));
              col[0] += lin.x; col[1] += lin.y; col[2] += lin.z;
              col[3] += axis.x; col[4] += axis.y; col[5] += axis.z;
            } else if (jt >= J_X_PRISM) {
              const int ax = jt - J_X_PRISM;
              col[0] += sign * C[ax]; col[1] += sign * C[4 + ax]; col[2] += sign * C[8 + ax];
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 6; r++) J[r * D + j] = col[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Forward kinematics with FOUR LANES PER POINT for the chain (the outputs the rollouts use: tool poses,
// spheres, cumulative transforms; the Jacobian / centre-of-mass outputs stay on the kernel above).
//
// What bounds this launch is not the 57 MB it writes but the serial work in front of the first store: every
// workgroup of the grid is resident at once, so the launch is (everything before the stores) + (the store
// stream at the rate of a plain fill, ~9 us), and a wavefront that has its SIMD to itself issues one
// instruction per ~7.5 cycles.  So the chain is laid out for the FEWEST INSTRUCTIONS PER LINK on the critical
// wavefront: lane c of a quad owns column c of the cumulative 3x4 (three registers).  A joint's local transform
// is F with two columns mixed (rotation about a coordinate axis) or one column added to the last (translation),
// i.e. column c of it is alpha * F[:,c] + beta * F[:,c'] with (alpha, beta, c') a function of (joint type, c)
// that is tabulated per link in LDS; column c of parent * local is then three FMAs per row whose parent
// operands come from the quad through DPP (fused into the FMA) -- ~30 instructions per link against ~110 for a
// lane that owns the whole matrix and ~30 for the 16-lane form above, which however keeps 4 of every 16 lanes
// idle and carries 4x the wavefronts.  sin/cos of the (point, joint) pairs are computed beforehand by all lanes
// (only the links that have a joint), and every global read of the block happens in one round trip up front.
constexpr int kFkPts = 64;  // points per workgroup (4 lanes each)

template <bool SPHERES, bool WRITE_CUMUL>
__global__ void __launch_bounds__(256) fk_forward_points_kernel(const FkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.nlinks, S = SPHERES ? a.nspheres : 0, D = a.njoints, T = a.n_tool_frames;
  float *cumul = smem;                                                 // [kFkPts][L][12]
  float2 *s_sc = reinterpret_cast<float2 *>(cumul + kFkPts * L * 12);  // [kFkPts][L] (sin, cos) | displacement
  float4 *s_col = reinterpret_cast<float4 *>(s_sc + kFkPts * L);       // [L][4][2] column tables (see above)
  float4 *s_sph = s_col + L * 8;                                       // [S]
  int *s_sph_link = reinterpret_cast<int *>(s_sph + S);                // [S]
  int4 *s_link = reinterpret_cast<int4 *>(s_sph_link + S + ((4 - (S & 3)) & 3));  // [L] (type, parent, joint, -)
  float2 *s_off = reinterpret_cast<float2 *>(s_link + L);              // [L] joint_offset (multiplier, bias)
  int *s_jointed = reinterpret_cast<int *>(s_off + L);                 // [L + 1] links with a joint, then their count
  float *s_q = reinterpret_cast<float *>(s_jointed + L + 1 + ((L + 1) & 1));  // [kFkPts][D]
  const int tid = threadIdx.x;
  const int pt0 = blockIdx.x * kFkPts;
  const int npts = min(kFkPts, a.n_points - pt0);
  const bool spheres_staged = SPHERES && a.num_envs <= 1;
  // ---- phase 0: everything the block reads from memory, in one round trip
  for (int i = tid; i < npts * D; i += blockDim.x) s_q[i] = a.q[(size_t)pt0 * D + i];
  for (int i = tid; i < L * 4; i += blockDim.x) {
    const int l = i >> 2, c = i & 3;
    fk_column_table_entry(s_col, i, a.joint_map_type, a.fixed_transform);
    if (c == 0) {
      s_link[l] = make_int4(a.joint_map_type[l], a.link_map[l], a.joint_map[l], 0);
      s_off[l] = make_float2(a.joint_offset[2 * l], a.joint_offset[2 * l + 1]);
    }
  }
  if (tid < kWave) {  // the links that have a joint, compacted (L <= 64 here: the LDS budget caps it lower)
    const bool jointed = tid < L && a.joint_map_type[tid < L ? tid : 0] != J_FIXED;
    const unsigned long long mask = __ballot(jointed);
    if (jointed) s_jointed[__popcll(mask & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) s_jointed[L] = __popcll(mask);
  }
  if (spheres_staged) {
    for (int s = tid; s < S; s += blockDim.x) {
      s_sph[s] = reinterpret_cast<const float4 *>(a.robot_spheres)[s];
      s_sph_link[s] = a.link_sphere_map[s];
    }
  }
  __syncthreads();
  // ---- phase 1: sin/cos of every (point, jointed link), all lanes, LDS to LDS
  {
    const int nj = s_jointed[L];
    for (int e = tid; e < npts * nj; e += blockDim.x) {
      const int lp = e / nj;
      const int l = s_jointed[e - lp * nj];
      const int4 tab = s_link[l];
      const float2 off = s_off[l];
      float sn, cs;
      joint_sincos(tab.x, s_q[lp * D + tab.z], off.x, off.y, &sn, &cs);
      s_sc[lp * L + l] = make_float2(sn, cs);
    }
  }
  __syncthreads();
  // ---- phase 2: the chain, one quad per point
  {
    const int c = tid & 3;
    const int me = (tid >> 2) < npts ? (tid >> 2) : 0;  // (idle quads shadow point 0: identical stores)
    fk_chain_quad(cumul + (size_t)me * L * 12, s_col, reinterpret_cast<const int *>(s_link) + 1, 4, L, c,
                  reinterpret_cast<const float *>(s_sc + me * L), 2);
  }
  __syncthreads();
  // ---- phase 3: outputs, all lanes, every stream contiguous over the workgroup's points
  if (WRITE_CUMUL) {
    const float4 *src = reinterpret_cast<const float4 *>(cumul);
    float4 *dst = reinterpret_cast<float4 *>(a.cumul_out + (size_t)pt0 * L * 12);
    for (int i = tid; i < npts * L * 3; i += blockDim.x) store_float4_streaming(dst + i, src[i]);
  }
  if (SPHERES) {
    // item e = (point, sphere) flat = the offset into the block's slice of the output; (point, sphere) are
    // stepped by 256 per round instead of divided
    float4 *out = reinterpret_cast<float4 *>(a.spheres_out) + (size_t)pt0 * S;
    const int step_p = (int)blockDim.x / S, step_s = (int)blockDim.x % S;
    int pt = tid / S, sp = tid - (tid / S) * S;
    for (int e = tid; e < npts * S; e += blockDim.x) {
      float4 rs;
      int link;
      if (spheres_staged) {
        rs = s_sph[sp];
        link = s_sph_link[sp];
      } else {
        const int env = a.env_query_idx[(pt0 + pt) / a.horizon];
        rs = reinterpret_cast<const float4 *>(a.robot_spheres)[(size_t)env * S + sp];
        link = a.link_sphere_map[sp];
      }
      store_float4_streaming(out + e, transform_sphere(cumul + ((size_t)pt * L + link) * 12, rs));
      sp += step_s; pt += step_p;
      if (sp >= S) { sp -= S; pt++; }
    }
  }
  // tool-frame poses (reference :270-301), quaternion written wxyz
  for (int e = tid; e < npts * T; e += blockDim.x) {
    const int pt = e / T, t = e - pt * T;
    const float *Cm = cumul + ((size_t)pt * L + a.tool_frame_map[t]) * 12;
    const float4 qx = quat_from_transform(Cm);
    reinterpret_cast<float4 *>(a.link_quat)[(size_t)pt0 * T + e] = make_float4(qx.w, qx.x, qx.y, qx.z);
    float *p = a.link_pos + ((size_t)pt0 * T + e) * 3;
    p[0] = Cm[3]; p[1] = Cm[7]; p[2] = Cm[11];
  }
}

// ------------------------------------------------------------------------------------------
// VJP.  reference kinematics_backward_kernel.cuh:27-157 (+ helpers, see file header).
// ------------------------------------------------------------------------------------------
struct FkBwdArgs {
  float *grad_q;
  const float *grad_link_pos;
  const float *grad_link_quat;
  const float *grad_spheres;
  const float *grad_spheres_b;
  const float *grad_com;
  const float *grad_jacobian;  // optional [n_points, n_tool_frames, 6, njoints]: VJP of the Jacobian output (dJ/dq)
  const float *batch_com;
  const float *cumul_in;
  const float *robot_spheres;
  const float *link_masses_com;
  const int8_t *joint_map_type;
  const int16_t *joint_map;
  const int16_t *tool_frame_map;
  const int16_t *link_sphere_map;
  const int16_t *link_chain_data;
  const int16_t *link_chain_offsets;
  const float *joint_offset;
  const int32_t *env_query_idx;
  int n_points, horizon, nspheres, num_envs, nlinks, njoints, n_tool_frames, dpad, chain_len;
  int stage_spheres;  // LDS was sized for the per-link form of the sphere pass
  int psum_rows;      // joint-gradient rows per point in LDS (16 = one per lane; fewer: lanes share rows through ds_add_f32)
};

template <bool COM>
__global__ void __launch_bounds__(256) fk_backward_kernel(const FkBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.nlinks, D = a.njoints;
  const int pts = blockDim.x / kFkLanes;
  float *cumul = smem;                          // [pts][L][12]
  const int R = a.psum_rows;
  float *psum_all = smem + pts * L * 12;        // [pts][R][dpad]
  int *s_chain_off = reinterpret_cast<int *>(psum_all + pts * R * a.dpad);  // [L+1]
  int *s_link_info = s_chain_off + (L + 1);                                        // [L]
  float *s_sign = reinterpret_cast<float *>(s_link_info + L);                      // [L]
  int *s_chain = reinterpret_cast<int *>(s_sign + L);                              // [C]
  // per-link form of the sphere pass (see below)
  const int S = a.nspheres;
  int *s_start = s_chain + a.chain_len;                                            // [L+1] first sphere of link l
  int *s_units = s_start + (L + 1);                                                // [kFkLanes + L] work units
  int *s_meta = s_units + (kFkLanes + L);                                          // [0] #units  [1] unsorted flag
  float4 *s_rs = reinterpret_cast<float4 *>(smem + (((s_meta + 2) - reinterpret_cast<int *>(smem) + 3) & ~3));  // [S]
  float4 *s_g = s_rs + S;                                                          // [pts][S] sphere gradients
  int *s_sph_link = reinterpret_cast<int *>(s_g + (size_t)pts * S);                // [S]
  const int tid = threadIdx.x;
  const int pt0 = blockIdx.x * pts;
  const int npts = min(pts, a.n_points - pt0);
  const bool staged = a.stage_spheres && S > 0;

  // ---- everything the block reads, requested in ONE burst: first the first round of every small table into
  // registers (clamped indices, so that no load hangs on a predicate), then the two big contiguous streams --
  // the saved cumulative transforms and (per-link form) the block's sphere gradients, both gradient streams
  // summed; the LDS stores follow.  Written as independent loops each table costs a dependent round trip.
  const int nt = blockDim.x;
  const int iL = min(tid, L - 1), iL1 = min(tid, L), iC = min(tid, a.chain_len - 1), iS = min(tid, S > 0 ? S - 1 : 0);
  const int r_off = a.link_chain_offsets[iL1];
  const int r_jt = a.joint_map_type[iL], r_jm = a.joint_map[iL];
  const float r_sign = a.joint_offset[2 * iL];
  const int r_chain = a.link_chain_data[iC];
  int r_lsm = 0;
  float4 r_rs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (staged) {
    r_lsm = a.link_sphere_map[iS];
    if (a.num_envs <= 1) r_rs = reinterpret_cast<const float4 *>(a.robot_spheres)[iS];
  }
  stage_float4<4>(reinterpret_cast<float4 *>(cumul), reinterpret_cast<const float4 *>(a.cumul_in + (size_t)pt0 * L * 12),
                  nullptr, npts * L * 3, tid, nt);
  if (staged)
    stage_float4<6>(s_g, reinterpret_cast<const float4 *>(a.grad_spheres) + (size_t)pt0 * S,
                    a.grad_spheres_b ? reinterpret_cast<const float4 *>(a.grad_spheres_b) + (size_t)pt0 * S : nullptr,
                    npts * S, tid, nt);
  s_chain_off[iL1] = r_off;
  s_link_info[iL] = (r_jt + 1) | ((r_jm < 0 ? 0 : r_jm) << 8);
  s_sign[iL] = r_sign;
  s_chain[iC] = r_chain;
  if (staged) {
    s_sph_link[iS] = r_lsm;
    if (a.num_envs <= 1) s_rs[iS] = r_rs;
    if (tid == 0) s_meta[1] = 0;
  }
  // (tables longer than the workgroup: the remaining rounds)
  for (int l = tid + nt; l <= L; l += nt) s_chain_off[l] = a.link_chain_offsets[l];
  for (int l = tid + nt; l < L; l += nt) {
    s_link_info[l] = ((int)a.joint_map_type[l] + 1) | ((int)(a.joint_map[l] < 0 ? 0 : a.joint_map[l]) << 8);
    s_sign[l] = a.joint_offset[2 * l];
  }
  for (int c = tid + nt; c < a.chain_len; c += nt) s_chain[c] = a.link_chain_data[c];
  if (staged)
    for (int sp = tid + nt; sp < S; sp += nt) {
      s_sph_link[sp] = a.link_sphere_map[sp];
      if (a.num_envs <= 1) s_rs[sp] = reinterpret_cast<const float4 *>(a.robot_spheres)[sp];
    }
  for (int i = tid; i < pts * R * a.dpad; i += nt) psum_all[i] = 0.0f;
  __syncthreads();
  // ---- work units of the per-link sphere pass.  With the spheres grouped by link (the order every robot file
  // lists them in) link l owns the run [start[l], start[l+1]); a run is cut into units of at most `cs` spheres so
  // that a point's 16 lanes get about one unit each whatever the distribution over the links (franka: 18 of 65 on
  // the hand).  A table that is not grouped by link takes the per-sphere form.
  if (staged) {
    for (int s = tid; s < S; s += blockDim.x) {
      const int lk = s_sph_link[s], prev = s > 0 ? s_sph_link[s - 1] : -1;
      if (lk < prev) s_meta[1] = 1;
      for (int l = prev + 1; l <= lk; l++) s_start[l] = s;
      if (s == S - 1)
        for (int l = lk + 1; l <= L; l++) s_start[l] = S;
    }
  }
  __syncthreads();
  const bool per_link = staged && s_meta[1] == 0;
  if (per_link && tid < kWave) {
    const int cs = (S + 11) / 12;
    int carry = 0;
    for (int l0 = 0; l0 < L; l0 += kWave) {  // (one pass unless the robot has more than 64 links)
      const int l = l0 + tid;
      const int b = l < L ? s_start[l] : 0, e = l < L ? s_start[l + 1] : 0;
      const int nu = (e - b + cs - 1) / cs;
      int incl = nu;  // inclusive scan over the wavefront
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if (tid >= off) incl += up;
      }
      const int at = carry + incl - nu;
      for (int k = 0; k < nu; k++) s_units[at + k] = l | ((b + k * cs) << 8) | (min(b + (k + 1) * cs, e) << 20);
      carry += __shfl(incl, kWave - 1, kWave);
    }
    if (tid == 0) s_meta[0] = carry;
  }
  __syncthreads();
  const BwdTables tb{s_chain, s_chain_off, s_link_info, s_sign};

  const int grp = tid / kFkLanes, lane = tid % kFkLanes;
  if (grp >= npts) return;
  const int n = pt0 + grp;
  const float *my_cumul = cumul + (size_t)grp * L * 12;
  float *psum = psum_all + ((size_t)grp * R + (lane & (R - 1))) * a.dpad;

  // ---- spheres (sparsity skip on zero gradient, reference :48-52)
  if (per_link) {
    // The reference pushes every sphere's gradient down the chain of its link (kinematics_backward_helper.cuh:
    // 62-98): S walks of ~depth joints per point.  The sum factors through the link: a lane adds up force and
    // torque of its unit's spheres (registers, no atomics) and walks the chain ONCE with the wrench.
    const int env = (a.num_envs > 1) ? a.env_query_idx[n / a.horizon] : 0;
    const float4 *rs_env = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)env * S;
    const float4 *g_pt = s_g + (size_t)grp * S;
    const int U = s_meta[0];
    for (int u = lane; u < U; u += kFkLanes) {
      const int unit = s_units[u];
      const int l = unit & 0xff, sb = (unit >> 8) & 0xfff, se = (unit >> 20) & 0xfff;
      const float *C = my_cumul + l * 12;
      f3 F = make_f3(0.f, 0.f, 0.f), T = make_f3(0.f, 0.f, 0.f);
      for (int sp = sb; sp < se; sp++) {
        const float4 g4 = g_pt[sp];
        if (g4.x == 0.0f && g4.y == 0.0f && g4.z == 0.0f) continue;
        const float4 pw = transform_sphere(C, a.num_envs > 1 ? rs_env[sp] : s_rs[sp]);
        const f3 g = make_f3(g4.x, g4.y, g4.z);
        F = F + g;
        T = T + cross(make_f3(pw.x, pw.y, pw.z), g);
      }
      if (F.x != 0.0f || F.y != 0.0f || F.z != 0.0f || T.x != 0.0f || T.y != 0.0f || T.z != 0.0f)
        chain_wrench_vjp(psum, my_cumul, tb, l, F, T);
    }
  } else if (a.nspheres > 0) {
    const int env = (a.num_envs > 1) ? a.env_query_idx[n / a.horizon] : 0;
    const float4 *rs = reinterpret_cast<const float4 *>(a.robot_spheres) + (size_t)env * a.nspheres;
    const float4 *ga = reinterpret_cast<const float4 *>(a.grad_spheres) + (size_t)n * a.nspheres;
    const float4 *gb = a.grad_spheres_b
                           ? reinterpret_cast<const float4 *>(a.grad_spheres_b) + (size_t)n * a.nspheres
                           : nullptr;
    // four spheres per lane and round: their gradient loads are requested together (clamped, unpredicated), so a
    // round is one memory round trip instead of four
    constexpr int U = 4;
    for (int s0 = lane; s0 < a.nspheres; s0 += kFkLanes * U) {
      float4 g4[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int sc = min(s0 + u * kFkLanes, a.nspheres - 1);
        g4[u] = ga[sc];
        if (gb) { const float4 h = gb[sc]; g4[u].x += h.x; g4[u].y += h.y; g4[u].z += h.z; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int sidx = s0 + u * kFkLanes;
        if (sidx >= a.nspheres || (g4[u].x == 0.0f && g4[u].y == 0.0f && g4[u].z == 0.0f)) continue;
        const int l = a.link_sphere_map[sidx];
        const float4 pw = transform_sphere(my_cumul + l * 12, rs[sidx]);
        chain_point_vjp(psum, my_cumul, tb, l, make_f3(pw.x, pw.y, pw.z), make_f3(g4[u].x, g4[u].y, g4[u].z));
      }
    }
  }
  // ---- tool frames: position + orientation (reference :102-183), chain split across lanes
  for (int t = 0; t < a.n_tool_frames; t++) {
    const float *gp = a.grad_link_pos + ((size_t)n * a.n_tool_frames + t) * 3;
    const float4 gq = reinterpret_cast<const float4 *>(a.grad_link_quat)[(size_t)n * a.n_tool_frames + t];
    const f3 g = make_f3(gp[0], gp[1], gp[2]);
    if (g.x == 0.f && g.y == 0.f && g.z == 0.f && gq.x == 0.f && gq.y == 0.f && gq.z == 0.f && gq.w == 0.f) continue;
    const int l = a.tool_frame_map[t];
    const float *C = my_cumul + l * 12;
    const float4 qx = quat_from_transform(C);
    const f3 pos = make_f3(C[3], C[7], C[11]);
    // omega = 0.5 * E(q)^T g  (reference quaternion_util.cuh:86-102; q xyzw, g wxyz)
    const float dqw = gq.x, dqx = gq.y, dqy = gq.z, dqz = gq.w;
    const f3 om = make_f3(0.5f * (-qx.x * dqw + qx.w * dqx + qx.z * dqy - qx.y * dqz),
                          0.5f * (-qx.y * dqw - qx.z * dqx + qx.w * dqy + qx.x * dqz),
                          0.5f * (-qx.z * dqw + qx.y * dqx - qx.x * dqy + qx.w * dqz));
    const int cs = s_chain_off[l], ce = s_chain_off[l + 1];
    for (int ci = cs + lane; ci < ce; ci += kFkLanes) {
      const int j = s_chain[ci];
      const int info = s_link_info[j];
      const int jt = (info & 0xff) - 1;
      if (jt < J_X_PRISM) continue;
      const float sign = s_sign[j];
      const float *Cj = my_cumul + j * 12;
      const int ax = jt >= J_X_ROT ? jt - J_X_ROT : jt;
      const f3 axis = make_f3(Cj[ax], Cj[4 + ax], Cj[8 + ax]);
      float r;
      if (jt >= J_X_ROT)
        r = dot(sign * g, cross(axis, pos - make_f3(Cj[3], Cj[7], Cj[11]))) + sign * dot(axis, om);
      else
        r = sign * dot(axis, g);
      atomicAdd(&psum[info >> 8], r);
    }
  }
  // ---- geometric-Jacobian output: grad_q[i] += sum_k <grad_J[:, k], dJ[:, k] / dq_i> (reference JAC_GRAD,
  // kinematics_jacobian_backward_helper.cuh:19-300).  Column k of tool frame t is (a_k x (p_ee - p_k), a_k) for a
  // revolute link k of the tool's chain (a = signed world axis) and (a_k, 0) for a prismatic one.  A joint i EARLIER in
  // the chain rotates / translates everything after it, one at or after k only moves the tool point:
  //   k rev, i rev,   i before k:  d a_k = a_i x a_k,  d Jv = (a_i x a_k) x (p_ee - p_k) + a_k x (a_i x (p_ee - p_k))
  //   k rev, i rev,   otherwise :  d Jv = a_k x (a_i x (p_ee - p_i))
  //   k rev, i prism, i after k :  d Jv = a_k x a_i          (before k: p_ee - p_k does not change)
  //   k prism, i rev, i before k:  d Jv = a_i x a_k
  // One chain position i per lane; every link of a (mimic) joint contributes, as in the forward Jacobian.
  if (a.grad_jacobian != nullptr) {
    for (int t = 0; t < a.n_tool_frames; t++) {
      const int tl = a.tool_frame_map[t];
      const float *E = my_cumul + tl * 12;
      const f3 pe = make_f3(E[3], E[7], E[11]);
      const int cs = s_chain_off[tl], ce = s_chain_off[tl + 1];
      const float *gJ = a.grad_jacobian + ((size_t)n * a.n_tool_frames + t) * 6 * D;
      for (int pi = cs + lane; pi < ce; pi += kFkLanes) {
        const int li = s_chain[pi];
        const int info_i = s_link_info[li];
        const int jt_i = (info_i & 0xff) - 1;
        if (jt_i < J_X_PRISM || li == 0) continue;
        const float *Ci = my_cumul + li * 12;
        const int ax_i = jt_i >= J_X_ROT ? jt_i - J_X_ROT : jt_i;
        const f3 a_i = s_sign[li] * make_f3(Ci[ax_i], Ci[4 + ax_i], Ci[8 + ax_i]);
        const f3 p_i = make_f3(Ci[3], Ci[7], Ci[11]);
        float r = 0.0f;
        for (int pk = cs; pk < ce; pk++) {
          const int lk = s_chain[pk];
          const int info_k = s_link_info[lk];
          const int jt_k = (info_k & 0xff) - 1;
          if (jt_k < J_X_PRISM || lk == 0) continue;
          const int k = info_k >> 8;
          const f3 gv = make_f3(gJ[0 * D + k], gJ[1 * D + k], gJ[2 * D + k]);
          const f3 gw = make_f3(gJ[3 * D + k], gJ[4 * D + k], gJ[5 * D + k]);
          if (gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gw.x == 0.f && gw.y == 0.f && gw.z == 0.f) continue;
          const float *Ck = my_cumul + lk * 12;
          const int ax_k = jt_k >= J_X_ROT ? jt_k - J_X_ROT : jt_k;
          const f3 a_k = s_sign[lk] * make_f3(Ck[ax_k], Ck[4 + ax_k], Ck[8 + ax_k]);
          const bool before = pi < pk;
          if (jt_k >= J_X_ROT) {
            const f3 p_k = make_f3(Ck[3], Ck[7], Ck[11]);
            if (jt_i >= J_X_ROT) {
              if (before) {
                const f3 dw = cross(a_i, a_k), ek = pe - p_k;
                r += dot(gw, dw) + dot(gv, cross(dw, ek) + cross(a_k, cross(a_i, ek)));
              } else {
                r += dot(gv, cross(a_k, cross(a_i, pe - p_i)));
              }
            } else if (!before) {
              r += dot(gv, cross(a_k, a_i));
            }
          } else if (jt_i >= J_X_ROT && before) {
            r += dot(gv, cross(a_i, a_k));
          }
        }
        if (r != 0.0f) atomicAdd(&psum[info_i >> 8], r);
      }
    }
  }
  // ---- centre of mass (reference :186-291)
  if (COM) {
    const float total_mass = a.batch_com[(size_t)n * 4 + 3];
    const float4 gc = reinterpret_cast<const float4 *>(a.grad_com)[n];
    if (total_mass > 0.0f && !(gc.x == 0.f && gc.y == 0.f && gc.z == 0.f)) {
      for (int l = lane; l < L; l += kFkLanes) {
        const float4 mc = reinterpret_cast<const float4 *>(a.link_masses_com)[l];
        if (mc.w <= 0.0f) continue;
        const f3 g = make_f3(gc.x * mc.w / total_mass, gc.y * mc.w / total_mass, gc.z * mc.w / total_mass);
        const float4 cw = transform_sphere(my_cumul + l * 12, mc);
        chain_point_vjp(psum, my_cumul, tb, l, make_f3(cw.x, cw.y, cw.z), g);
      }
    }
  }
  // ---- 16-lane reduction of the per-lane rows.  All 16 lanes of a point sit in one wave and
  // DS operations of a wave complete in order, so a wave-level fence suffices.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float *rows = psum_all + (size_t)grp * R * a.dpad;
  for (int j = lane; j < D; j += kFkLanes) {
    float acc = 0.0f;
    for (int k = 0; k < R; k++) acc += rows[k * a.dpad + j];
    a.grad_q[(size_t)n * D + j] = acc;
  }
}

// ---------------------------------------------------------------- host launchers
static int fk_points_per_block(int nlinks) {
  // LDS per point: forward 28*L floats, backward 12*L + 16*dpad; keep a block <= ~48 KiB so that
  // >= 3 blocks share a CU's 160 KiB.
  if (nlinks <= 24) return 16;
  if (nlinks <= 64) return 8;
  return 4;
}

template <bool S, bool J, bool C>
static void launch_fk(const FkArgs &a, bool write_cumul, int blocks, int threads, size_t lds, hipStream_t st) {
  if (write_cumul) hipLaunchKernelGGL((fk_forward_kernel<S, J, C, true>), dim3(blocks), dim3(threads), lds, st, a);
  else hipLaunchKernelGGL((fk_forward_kernel<S, J, C, false>), dim3(blocks), dim3(threads), lds, st, a);
}

static int fk_forward_dispatch(const FkArgs &a, bool spheres, bool jac, bool com, bool write_cumul,
                               hipStream_t st, const char *what) {
  CUROBO_REQUIRE(a.nlinks >= 1 && a.nlinks <= 128, "%s: num_links=%d out of range [1,128]", what, a.nlinks);
  CUROBO_REQUIRE(a.n_points >= 0 && a.horizon >= 1, "%s: bad batch_size/horizon", what);
  CUROBO_REQUIRE(a.njoints >= 1, "%s: n_joints must be >= 1", what);
  if (a.n_points == 0) return CUROBO_HIP_OK;
  static const bool grouped_only = getenv("CUROBO_HIP_FK_GROUPED") != nullptr;
  const size_t lds_pts = (size_t)kFkPts * a.nlinks * (12 * sizeof(float) + sizeof(float2)) +
                         (size_t)a.nlinks * (8 * sizeof(float4) + sizeof(int4) + sizeof(float2) + sizeof(int)) + 16 +
                         (size_t)kFkPts * a.njoints * sizeof(float) +
                         (spheres ? (size_t)(a.nspheres + 4) * (sizeof(float4) + sizeof(int)) : 0);
  if (!jac && !com && !grouped_only && lds_pts <= 80 * 1024 && a.nlinks <= kWave) {  // (two workgroups per CU)
    const int nblk = ceil_div(a.n_points, kFkPts);
    if (spheres && write_cumul) hipLaunchKernelGGL((fk_forward_points_kernel<true, true>), dim3(nblk), dim3(256), lds_pts, st, a);
    else if (spheres) hipLaunchKernelGGL((fk_forward_points_kernel<true, false>), dim3(nblk), dim3(256), lds_pts, st, a);
    else if (write_cumul) hipLaunchKernelGGL((fk_forward_points_kernel<false, true>), dim3(nblk), dim3(256), lds_pts, st, a);
    else hipLaunchKernelGGL((fk_forward_points_kernel<false, false>), dim3(nblk), dim3(256), lds_pts, st, a);
    return check_launch(what, st);
  }
  static const int env_pts = getenv("CUROBO_FK_FWD_PTS") ? atoi(getenv("CUROBO_FK_FWD_PTS")) : 0;
  auto lds_for = [&](int p) {
    return (size_t)p * a.nlinks * kFkLinkStride * sizeof(float) + (size_t)a.nlinks * sizeof(int) +
           (spheres ? (size_t)a.nspheres * (sizeof(float4) + sizeof(int)) : 0);
  };
  int pts = fk_points_per_block(a.nlinks);
  // every workgroup stages the sphere table: with many spheres (G1: 13.5 KB) twice the points per workgroup hold more
  // wavefronts per CU in the same LDS (two workgroups of 16 points instead of three of 8: 151 -> 145 us at the C4 size)
  if (spheres && pts == 8 && lds_for(16) <= 80 * 1024) pts = 16;
  if (env_pts > 0) pts = env_pts;
  const int threads = pts * kFkLanes;
  const int blocks = ceil_div(a.n_points, pts);
  const size_t lds = lds_for(pts);
  const int key = (spheres ? 1 : 0) | (jac ? 2 : 0) | (com ? 4 : 0);
  switch (key) {
    case 0: launch_fk<false, false, false>(a, write_cumul, blocks, threads, lds, st); break;
    case 1: launch_fk<true, false, false>(a, write_cumul, blocks, threads, lds, st); break;
    case 2: launch_fk<false, true, false>(a, write_cumul, blocks, threads, lds, st); break;
    case 3: launch_fk<true, true, false>(a, write_cumul, blocks, threads, lds, st); break;
    case 4: launch_fk<false, false, true>(a, write_cumul, blocks, threads, lds, st); break;
    case 5: launch_fk<true, false, true>(a, write_cumul, blocks, threads, lds, st); break;
    case 6: launch_fk<false, true, true>(a, write_cumul, blocks, threads, lds, st); break;
    default: launch_fk<true, true, true>(a, write_cumul, blocks, threads, lds, st); break;
  }
  return check_launch(what, st);
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_launch_kinematics_forward(
    float *link_pos, float *link_quat, float *batch_center_of_mass, float *global_cumul_mat,
    const float *joint_vec, const float *fixed_transform, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map,
    const int16_t *tool_frame_map, const float *joint_offset_map, int batch_size, int horizon,
    int n_joints, int num_links, int n_tool_frames, int compute_com, curobo_hip_stream_t stream) {
  FkArgs a{};
  a.link_pos = link_pos; a.link_quat = link_quat; a.com_out = batch_center_of_mass;
  a.cumul_out = global_cumul_mat; a.q = joint_vec; a.fixed_transform = fixed_transform;
  a.link_masses_com = link_masses_com; a.joint_map_type = joint_map_type; a.joint_map = joint_map;
  a.link_map = link_map; a.tool_frame_map = tool_frame_map; a.joint_offset = joint_offset_map;
  a.n_points = batch_size; a.horizon = horizon; a.nspheres = 0; a.num_envs = 1;
  a.nlinks = num_links; a.njoints = n_joints; a.n_tool_frames = n_tool_frames;
  return fk_forward_dispatch(a, false, false, compute_com != 0, global_cumul_mat != nullptr,
                             (hipStream_t)stream, "launch_kinematics_forward");
}

CUROBO_EXPORT int curobo_hip_launch_kinematics_forward_spheres(
    float *link_pos, float *link_quat, float *batch_robot_spheres, float *batch_center_of_mass,
    float *global_cumul_mat, const float *joint_vec, const float *fixed_transform,
    const float *robot_spheres, const float *link_masses_com, const int8_t *joint_map_type,
    const int16_t *joint_map, const int16_t *link_map, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const float *joint_offset_map, const int32_t *env_query_idx,
    int num_envs, int batch_size, int horizon, int n_joints, int num_spheres, int num_links,
    int n_tool_frames, int write_global_cumul, int compute_com, curobo_hip_stream_t stream) {
  FkArgs a{};
  a.link_pos = link_pos; a.link_quat = link_quat; a.spheres_out = batch_robot_spheres;
  a.com_out = batch_center_of_mass; a.cumul_out = global_cumul_mat; a.q = joint_vec;
  a.fixed_transform = fixed_transform; a.robot_spheres = robot_spheres;
  a.link_masses_com = link_masses_com; a.joint_map_type = joint_map_type; a.joint_map = joint_map;
  a.link_map = link_map; a.tool_frame_map = tool_frame_map; a.link_sphere_map = link_sphere_map;
  a.joint_offset = joint_offset_map; a.env_query_idx = env_query_idx;
  a.n_points = batch_size; a.horizon = horizon; a.nspheres = num_spheres; a.num_envs = num_envs;
  a.nlinks = num_links; a.njoints = n_joints; a.n_tool_frames = n_tool_frames;
  return fk_forward_dispatch(a, num_spheres > 0, false, compute_com != 0, write_global_cumul != 0,
                             (hipStream_t)stream, "launch_kinematics_forward_spheres");
}

CUROBO_EXPORT int curobo_hip_launch_kinematics_forward_spheres_jacobian(
    float *link_pos, float *link_quat, float *batch_robot_spheres, float *batch_center_of_mass,
    float *batch_jacobian, float *global_cumul_mat, const float *joint_vec,
    const float *fixed_transform, const float *robot_spheres, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map,
    const int16_t *tool_frame_map, const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const int32_t *env_query_idx, int num_envs, int batch_size,
    int horizon, int n_joints, int num_spheres, int num_links, int n_tool_frames,
    int write_global_cumul, int compute_com, curobo_hip_stream_t stream) {
  FkArgs a{};
  a.link_pos = link_pos; a.link_quat = link_quat; a.spheres_out = batch_robot_spheres;
  a.com_out = batch_center_of_mass; a.jacobian_out = batch_jacobian; a.cumul_out = global_cumul_mat;
  a.q = joint_vec; a.fixed_transform = fixed_transform; a.robot_spheres = robot_spheres;
  a.link_masses_com = link_masses_com; a.joint_map_type = joint_map_type; a.joint_map = joint_map;
  a.link_map = link_map; a.tool_frame_map = tool_frame_map; a.link_sphere_map = link_sphere_map;
  a.link_chain_data = link_chain_data; a.link_chain_offsets = link_chain_offsets;
  a.joint_links_data = joint_links_data; a.joint_links_offsets = joint_links_offsets;
  a.joint_affects_endeffector = joint_affects_endeffector; a.joint_offset = joint_offset_map;
  a.env_query_idx = env_query_idx;
  a.n_points = batch_size; a.horizon = horizon; a.nspheres = num_spheres; a.num_envs = num_envs;
  a.nlinks = num_links; a.njoints = n_joints; a.n_tool_frames = n_tool_frames;
  return fk_forward_dispatch(a, num_spheres > 0, true, compute_com != 0, write_global_cumul != 0,
                             (hipStream_t)stream, "launch_kinematics_forward_spheres_jacobian");
}

CUROBO_EXPORT int curobo_hip_launch_kinematics_backward(
    float *grad_out, const float *grad_nlinks_pos, const float *grad_nlinks_quat,
    const float *grad_spheres, const float *grad_spheres_b, const float *grad_center_of_mass,
    const float *batch_center_of_mass, const float *grad_jacobian, const float *global_cumul_mat,
    const float *robot_spheres, const float *link_masses_com, const int16_t *link_map,
    const int16_t *joint_map, const int8_t *joint_map_type, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const int16_t *joint_links_data,
    const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const int32_t *env_query_idx, int num_envs, int batch_size,
    int horizon, int n_joints, int num_spheres, int num_links, int n_tool_frames, int link_chain_len,
    int compute_com, int compute_jacobian_grad, curobo_hip_stream_t stream) {
  (void)link_map; (void)joint_links_data; (void)joint_links_offsets; (void)joint_affects_endeffector;
  const char *what = "launch_kinematics_backward";
  CUROBO_REQUIRE(!compute_jacobian_grad || grad_jacobian, "%s: compute_jacobian_grad needs grad_jacobian", what);
  CUROBO_REQUIRE(num_links >= 1 && num_links <= 128, "%s: num_links=%d out of range [1,128]", what, num_links);
  CUROBO_REQUIRE(n_joints >= 1 && n_joints <= 1024, "%s: n_joints=%d out of range", what, n_joints);
  CUROBO_REQUIRE(link_chain_len >= 1 && link_chain_len <= 16384, "%s: link_chain_len=%d out of range", what, link_chain_len);
  CUROBO_REQUIRE(((uintptr_t)grad_nlinks_quat & 15) == 0, "%s: grad_nlinks_quat is not aligned to 16 bytes", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  FkBwdArgs a{};
  a.grad_q = grad_out; a.grad_link_pos = grad_nlinks_pos; a.grad_link_quat = grad_nlinks_quat;
  a.grad_spheres = grad_spheres; a.grad_spheres_b = grad_spheres_b; a.grad_com = grad_center_of_mass;
  a.grad_jacobian = compute_jacobian_grad ? grad_jacobian : nullptr;
  a.batch_com = batch_center_of_mass; a.cumul_in = global_cumul_mat; a.robot_spheres = robot_spheres;
  a.link_masses_com = link_masses_com; a.joint_map_type = joint_map_type; a.joint_map = joint_map;
  a.tool_frame_map = tool_frame_map; a.link_sphere_map = link_sphere_map;
  a.link_chain_data = link_chain_data; a.link_chain_offsets = link_chain_offsets;
  a.joint_offset = joint_offset_map; a.env_query_idx = env_query_idx;
  a.n_points = batch_size; a.horizon = horizon; a.nspheres = grad_spheres ? num_spheres : 0;
  a.num_envs = num_envs; a.nlinks = num_links; a.njoints = n_joints; a.n_tool_frames = n_tool_frames;
  a.dpad = n_joints | 1;  // odd row stride: conflict-free column reads
  a.chain_len = link_chain_len;
  // 8 points per workgroup where 16 would fit: twice the workgroups, whose load and arithmetic phases then overlap
  // a little more (24.6 -> 23.5 us at 32 k points)
  int pts = max(4, fk_points_per_block(num_links) / 2);
  static const int env_rows = getenv("CUROBO_FK_BWD_ROWS") ? atoi(getenv("CUROBO_FK_BWD_ROWS")) : 0;
  static const int env_pts = getenv("CUROBO_FK_BWD_PTS") ? atoi(getenv("CUROBO_FK_BWD_PTS")) : 0;
  // One gradient row per lane costs 16 * dof floats of LDS per point: with many joints that, not the wavefront slots, is what
  // limits the points a CU holds (Unitree G1, 49 joints: 6.1 KB per point = 6.5 wavefronts per CU).  Four rows shared by the
  // lanes through ds_add_f32 make it 3.5 KB (G1 at the C4 size, sparse sphere gradient + four tool frames: 314 -> 194 us);
  // robots with few joints keep a private row per lane (no LDS pressure, no shared-address adds).
  a.psum_rows = env_rows > 0 ? env_rows : (a.dpad > 16 ? 4 : kFkLanes);
  if (a.psum_rows < kFkLanes) pts = fk_points_per_block(num_links);
  if (env_pts > 0) pts = env_pts;
  size_t lds = 0;
  for (;;) {
    lds = ((size_t)pts * num_links * 12 + (size_t)pts * a.psum_rows * a.dpad + 3 * (size_t)num_links + 1 +
           (size_t)a.chain_len) * sizeof(float);
    if (lds <= 60 * 1024 || pts == 4) break;
    pts /= 2;
  }
  {  // + the tables and the gradient slab of the per-link sphere pass, when they leave >= 3 workgroups per CU
    const size_t S = (size_t)a.nspheres;
    const size_t extra = ((size_t)2 * num_links + kFkLanes + 8) * sizeof(int) + (S + 4) * (sizeof(int) + sizeof(float4)) +
                         (size_t)pts * S * sizeof(float4) + 16;
    a.stage_spheres = S > 0 && S < 4096 && num_links < 256 && lds + extra <= 48 * 1024;
    if (a.stage_spheres) lds += extra;
  }
  static const bool per_sphere_only = getenv("CUROBO_HIP_FK_BWD_PER_SPHERE") != nullptr;
  if (per_sphere_only) a.stage_spheres = 0;
  CUROBO_REQUIRE(lds <= 64 * 1024, "%s: robot too large for the LDS tiling (%zu bytes)", what, lds);
  const int threads = pts * kFkLanes;
  const int blocks = ceil_div(batch_size, threads / kFkLanes);
  if (compute_com)
    hipLaunchKernelGGL((fk_backward_kernel<true>), dim3(blocks), dim3(threads), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((fk_backward_kernel<false>), dim3(blocks), dim3(threads), lds, (hipStream_t)stream, a);
  return check_launch(what, (hipStream_t)stream);
}
