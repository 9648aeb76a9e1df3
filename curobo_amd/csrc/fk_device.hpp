// fk_device.hpp -- device functions of forward kinematics and its VJP shared by kinematics.hip
// and the fused rollout kernels.  Reference: kernels/kinematics/kinematics_forward_helper.cuh
// :316-512, kinematics_backward_helper.cuh:14-183, kinematics_joint_util.cuh:13-66.
#pragma once
#include "common.hpp"

namespace curobo_hip {

constexpr int kFkLanes = 16;  // lanes per point

// reference kinematics_forward_helper.cuh:316-393 / kinematics_util.cuh:62-74, in two steps so that the
// transcendental can be computed by other lanes than the ones that build the matrix:
//   joint_sincos: (s, c) of the joint angle (revolute), or s = the displacement (prismatic)
//   local_transform_from_sincos: the local 3x4 of one (point, link) column-major, each column padded to 4 floats
__device__ __forceinline__ void joint_sincos(int j_type, float q_val, float off_mul, float off_add, float *s, float *c) {
  *s = 0.0f; *c = 1.0f;
  if (j_type != J_FIXED) {
    const float angle = off_mul * q_val + off_add;
    if (j_type <= J_Z_PRISM) *s = angle;
    else sincos_bounded(angle, s, c);
  }
}

__device__ __forceinline__ void local_transform_from_sincos(float *__restrict__ dst, const float *__restrict__ F,
                                                            int j_type, float s, float c) {
  const float f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3];
  const float f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7];
  const float f8 = F[8], f9 = F[9], f10 = F[10], f11 = F[11];
  float4 c0 = make_float4(f0, f4, f8, 0.0f);
  float4 c1 = make_float4(f1, f5, f9, 0.0f);
  float4 c2 = make_float4(f2, f6, f10, 0.0f);
  float4 c3 = make_float4(f3, f7, f11, 0.0f);
  if (j_type != J_FIXED) {
    if (j_type <= J_Z_PRISM) {
      const float angle = s;
      c3.x = f3 + (j_type == J_X_PRISM ? f0 : (j_type == J_Y_PRISM ? f1 : f2)) * angle;
      c3.y = f7 + (j_type == J_X_PRISM ? f4 : (j_type == J_Y_PRISM ? f5 : f6)) * angle;
      c3.z = f11 + (j_type == J_X_PRISM ? f8 : (j_type == J_Y_PRISM ? f9 : f10)) * angle;
    } else {
      const int xyz = j_type - J_X_ROT;
      const float is_x = xyz == 0 ? 1.0f : 0.0f;
      const float is_y = xyz == 1 ? 1.0f : 0.0f;
      const float is_z = xyz == 2 ? 1.0f : 0.0f;
      const float s0 = is_x + c * (is_y + is_z);
      const float s1 = is_y + c * (is_x + is_z);
      const float s2 = is_z + c * (is_x + is_y);
      c0 = make_float4(f0 * s0 + s * (is_z * f1 - is_y * f2), f4 * s0 + s * (is_z * f5 - is_y * f6),
                       f8 * s0 + s * (is_z * f9 - is_y * f10), 0.0f);
      c1 = make_float4(f1 * s1 + s * (is_x * f2 - is_z * f0), f5 * s1 + s * (is_x * f6 - is_z * f4),
                       f9 * s1 + s * (is_x * f10 - is_z * f8), 0.0f);
      c2 = make_float4(f2 * s2 + s * (is_y * f0 - is_x * f1), f6 * s2 + s * (is_y * f4 - is_x * f5),
                       f10 * s2 + s * (is_y * f8 - is_x * f9), 0.0f);
    }
  }
  float4 *d4 = reinterpret_cast<float4 *>(dst);
  d4[0] = c0; d4[1] = c1; d4[2] = c2; d4[3] = c3;
}

__device__ __forceinline__ void local_transform_colmajor(float *__restrict__ dst, const float *__restrict__ F,
                                                        int j_type, float q_val, float off_mul,
                                                        float off_add) {
  float s, c;
  joint_sincos(j_type, q_val, off_mul, off_add, &s, &c);
  local_transform_from_sincos(dst, F, j_type, s, c);
}

// ds_read_b32 + wait as one opaque unit (LDS operations of a wave execute in issue order, so the read sees
// every earlier ds_write of the wave)
__device__ __forceinline__ float lds_read_f32_now(const float *p) {
  float v;
  const uint32_t a = (uint32_t)(uintptr_t)p;  // LDS pointers: the low 32 bits are the LDS byte address
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// Serial chain of ONE point, executed by its 16-lane group without any barrier (lane 4r+c owns
// element (r, c) of every cumulative 3x4; a lane only re-reads entries it wrote itself, the three
// rotation entries of its row come from its DPP quad).  `local` holds the point's column-major
// padded local transforms [L][16], `cumul` receives [L][12] row-major, `parent` = link_map.
__device__ __forceinline__ void fk_chain_16(float *__restrict__ cumul, const float *__restrict__ local,
                                            const int *__restrict__ parent, const float *__restrict__ fixed_transform,
                                            int L, int lane) {
  const int c = lane & 3;
  const bool owner = lane < 12;
  float cur = owner ? fixed_transform[lane] : 0.0f;  // base link: reference :467-485
  if (owner) cumul[lane] = cur;
  const float *my_local = local + c * 4;
  for (int l = 1; l < L; l++) {
    const int par = __builtin_amdgcn_readfirstlane(parent[l]);
    float p = cur;
    if (par != l - 1) p = lds_read_f32_now(cumul + par * 12 + lane);  // see fk_chain_16_multi
    p = owner ? p : 0.0f;
    const float a0 = quad_bcast<0>(p), a1 = quad_bcast<1>(p), a2 = quad_bcast<2>(p), a3 = quad_bcast<3>(p);
    const float4 m = *reinterpret_cast<const float4 *>(my_local + l * 16);
    cur = a0 * m.x + a1 * m.y + a2 * m.z + (c == 3 ? a3 : 0.0f);
    if (owner) cumul[l * 12 + lane] = cur;
  }
}

// fk_chain_16 with the cumulative transform of link l written OVER the local transform of link l (both at
// buf + l * 16; the 3x4 takes the first 12 floats).  Every lane of the group has read its column of local[l]
// before any lane writes cumul[l] (one wavefront, LDS operations execute in issue order), and nothing reads
// local[l] again: the kernel needs 16 instead of 28 floats of LDS per (point, link).  No __restrict__ here: the
// read of local[l] must stay ahead of the write to the same words.
__device__ __forceinline__ void fk_chain_16_inplace(float *buf, const int *__restrict__ parent,
                                                    const float *__restrict__ fixed_transform, int L, int lane) {
  const int c = lane & 3;
  const bool owner = lane < 12;
  float cur = owner ? fixed_transform[lane] : 0.0f;
  if (owner) buf[lane] = cur;
  for (int l = 1; l < L; l++) {
    const int par = __builtin_amdgcn_readfirstlane(parent[l]);
    const float4 m = *reinterpret_cast<const float4 *>(buf + l * 16 + c * 4);
    float p = cur;
    if (par != l - 1) p = lds_read_f32_now(buf + par * 16 + lane);
    p = owner ? p : 0.0f;
    const float a0 = quad_bcast<0>(p), a1 = quad_bcast<1>(p), a2 = quad_bcast<2>(p), a3 = quad_bcast<3>(p);
    cur = a0 * m.x + a1 * m.y + a2 * m.z + (c == 3 ? a3 : 0.0f);
    if (owner) buf[l * 16 + lane] = cur;
  }
}

// what column c of the local transform of a joint of type jt is made of (fk_chain_quad): bit 0: alpha = cos (else
// 1); bit 1: beta != 0; bit 2: beta negative; bits 3..4: c'
__device__ __forceinline__ int local_column_code(int jt, int c) {
  switch (jt) {
    case J_X_PRISM: return c == 3 ? (2 | (0 << 3)) : 0;
    case J_Y_PRISM: return c == 3 ? (2 | (1 << 3)) : 0;
    case J_Z_PRISM: return c == 3 ? (2 | (2 << 3)) : 0;
    case J_X_ROT: return c == 1 ? (1 | 2 | (2 << 3)) : (c == 2 ? (1 | 2 | 4 | (1 << 3)) : 0);
    case J_Y_ROT: return c == 0 ? (1 | 2 | 4 | (2 << 3)) : (c == 2 ? (1 | 2 | (0 << 3)) : 0);
    case J_Z_ROT: return c == 0 ? (1 | 2 | (1 << 3)) : (c == 1 ? (1 | 2 | 4 | (0 << 3)) : 0);
    default: return 0;
  }
}

// Serial chain of ONE point on ONE QUAD: lane c owns column c of the cumulative 3x4 (three registers).  A joint's
// local transform is F with two columns mixed (rotation about a coordinate axis) or one column added to the last
// (translation), i.e. column c of it is alpha * F[:,c] + beta * F[:,c'] with (alpha, beta, c') a function of
// (joint type, c): col_tab[(l * 4 + c) * 2] = (F[:,c], code as bits), [.. + 1] = (F[:,c'], -).  Column c of
// parent * local is three FMAs per row whose parent operands come from the quad through DPP: ~30 instructions
// per link.  `cumul_pt` = the point's [L][12] row-major output, (sin, cos) of link l at sc[l * sc_stride]
// (may alias cumul_pt[l * 12]: read one link ahead of the write), parent of link l at parent[l * pstride].
template <int UNROLL = 4>
__device__ __forceinline__ void fk_chain_quad(float *cumul_pt, const float4 *__restrict__ col_tab, const int *__restrict__ parent,
                                              int pstride, int L, int c, const float *sc, int sc_stride) {
  float *mine = cumul_pt + c;
  float P0 = col_tab[c * 2].x, P1 = col_tab[c * 2].y, P2 = col_tab[c * 2].z;  // base link: reference :467-485
  const float last = c == 3 ? 1.0f : 0.0f;
  mine[0] = P0; mine[4] = P1; mine[8] = P2;
  // UNROLL links per round: their operands (two table rows, sin / cos, parent) are requested together, so a round
  // costs ONE LDS round trip plus its arithmetic.  Link by link every step is a dependent round trip -- cheap on an
  // idle CU, several hundred cycles when a co-resident workgroup is hammering the LDS.
  for (int l0 = 1; l0 < L; l0 += UNROLL) {
    float4 A[UNROLL], B[UNROLL];
    float2 sc_l[UNROLL];
    int par[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int l = l0 + u < L ? l0 + u : L - 1;
      A[u] = col_tab[(l * 4 + c) * 2]; B[u] = col_tab[(l * 4 + c) * 2 + 1];
      sc_l[u] = *reinterpret_cast<const float2 *>(sc + l * sc_stride);
      par[u] = parent[l * pstride];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int l = l0 + u;
      if (l < L) {
        const int code = __builtin_bit_cast(int, A[u].w);
        const float alpha = (code & 1) ? sc_l[u].y : 1.0f;
        const float beta = (code & 2) ? ((code & 4) ? -sc_l[u].x : sc_l[u].x) : 0.0f;
        const float M0 = A[u].x * alpha + beta * B[u].x, M1 = A[u].y * alpha + beta * B[u].y, M2 = A[u].z * alpha + beta * B[u].z;
        const int pl = __builtin_amdgcn_readfirstlane(par[u]);
        if (pl != l - 1) {  // a branch of the tree: the quad's own earlier result (LDS operations stay in order)
          P0 = mine[pl * 12]; P1 = mine[pl * 12 + 4]; P2 = mine[pl * 12 + 8];
        }
        const float C0 = quad_bcast<0>(P0) * M0 + quad_bcast<1>(P0) * M1 + quad_bcast<2>(P0) * M2 + last * quad_bcast<3>(P0);
        const float C1 = quad_bcast<0>(P1) * M0 + quad_bcast<1>(P1) * M1 + quad_bcast<2>(P1) * M2 + last * quad_bcast<3>(P1);
        const float C2 = quad_bcast<0>(P2) * M0 + quad_bcast<1>(P2) * M1 + quad_bcast<2>(P2) * M2 + last * quad_bcast<3>(P2);
        mine[l * 12] = C0; mine[l * 12 + 4] = C1; mine[l * 12 + 8] = C2;
        P0 = C0; P1 = C1; P2 = C2;
      }
    }
  }
}

// the column table of fk_chain_quad from the global robot tables: entry i = (link i / 4, column i % 4)
__device__ __forceinline__ void fk_column_table_entry(float4 *col_tab, int i, const int8_t *joint_map_type,
                                                      const float *fixed_transform) {
  const int l = i >> 2, c = i & 3;
  const int code = local_column_code(joint_map_type[l], c);
  const float *F = fixed_transform + l * 12;
  const int c2 = code >> 3;
  col_tab[i * 2] = make_float4(F[c], F[4 + c], F[8 + c], __builtin_bit_cast(float, code));
  col_tab[i * 2 + 1] = make_float4(F[c2], F[4 + c2], F[8 + c2], 0.0f);
}

// The same chain for N independent points in one instruction stream, software-pipelined: the walk
// is a string of dependent steps, so everything that does not depend on the previous link (the
// local transforms and parent indices of the next UNROLL links) is fetched up front and the
// dependent part of a step is 4 DPP broadcasts + 3 FMAs; a second point rides in the first one's
// latency shadow.
template <int N, int UNROLL = 4>
__device__ __forceinline__ void fk_chain_16_multi(float *const (&cumul)[N], const float *const (&local)[N],
                                                  const int *__restrict__ parent, const float *__restrict__ fixed_transform,
                                                  int L, int lane) {
  const int c = lane & 3;
  const bool owner = lane < 12;
  float cur[N];
  const float base = owner ? fixed_transform[lane] : 0.0f;
#pragma unroll
  for (int n = 0; n < N; n++) {
    cur[n] = base;
    if (owner) cumul[n][lane] = base;
  }
  for (int l0 = 1; l0 < L; l0 += UNROLL) {
    float4 m[UNROLL][N];
    int par[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int l = l0 + u < L ? l0 + u : L - 1;
      // the parent table is the same for every lane: a scalar value makes `par != l - 1` a scalar branch
      // (no exec-mask round trip, no VGPR compare on the dependent path of the chain)
      par[u] = __builtin_amdgcn_readfirstlane(parent[l]);
#pragma unroll
      for (int n = 0; n < N; n++) m[u][n] = *reinterpret_cast<const float4 *>(local[n] + c * 4 + l * 16);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int l = l0 + u;
      if (l < L) {
#pragma unroll
        for (int n = 0; n < N; n++) {
          float p = cur[n];
          // A parent that is not the previous link (tree branches) is re-read from LDS.  The read sits in an
          // opaque asm block with its own wait so that it stays behind a real scalar branch: as plain C++ the
          // compiler if-converts it into an exec-masked load followed by an UNCONDITIONAL s_waitcnt
          // lgkmcnt(0), and every step of the chain then also waits for the previous step's ds_write
          // (LDS returns in order) -- a full LDS round trip per link on the dependent path.
          if (par[u] != l - 1) p = lds_read_f32_now(cumul[n] + par[u] * 12 + lane);
          p = owner ? p : 0.0f;
          const float a0 = quad_bcast<0>(p), a1 = quad_bcast<1>(p), a2 = quad_bcast<2>(p), a3 = quad_bcast<3>(p);
          cur[n] = a0 * m[u][n].x + a1 * m[u][n].y + a2 * m[u][n].z + (c == 3 ? a3 : 0.0f);
          if (owner) cumul[n][l * 12 + lane] = cur[n];
        }
      }
    }
  }
}

// Small robot tables staged in LDS once per workgroup (the chain walk is a pointer chase; from
// global memory every step is a dependent L1/L2 round trip).
struct BwdTables {
  const int *chain;      // [C]   link indices, CSR data
  const int *chain_off;  // [L+1] CSR offsets
  const int *link_info;  // [L]   (joint_type + 1) | joint_index << 8   (joint_type in [-1,5])
  const float *sign;     // [L]   joint_offset[2*l] (axis sign x mimic multiplier)
};

// gradient of one world point p with cost gradient g, pushed down the chain of link `l`
// (reference kinematics_backward_helper.cuh:62-98, kinematics_joint_util.cuh:13-66)
__device__ __forceinline__ void chain_point_vjp(float *__restrict__ psum, const float *__restrict__ cumul,
                                                const BwdTables &t, int l, f3 p, f3 g) {
  const int cs = t.chain_off[l];
  for (int ci = t.chain_off[l + 1] - 1; ci >= cs; ci--) {
    const int j = t.chain[ci];
    const int info = t.link_info[j];
    const int jt = (info & 0xff) - 1;
    if (jt < J_X_PRISM) continue;
    const float sign = t.sign[j];
    const float *C = cumul + j * 12;
    const int ax = jt >= J_X_ROT ? jt - J_X_ROT : jt;
    const f3 axis = make_f3(C[ax], C[4 + ax], C[8 + ax]);
    float r;
    if (jt >= J_X_ROT) r = dot(sign * g, cross(axis, p - make_f3(C[3], C[7], C[11])));
    else r = sign * dot(axis, g);
    atomicAdd(&psum[info >> 8], r);  // own LDS row: ds_add_f32, never contended
  }
}

// the same for a WRENCH on link `l`: force F and torque T about the world origin (the sum of g and of p x g over
// points p rigidly attached to the link).  sum_p <sign g, axis x (p - o)> = sign <axis, T - o x F>: one walk per
// link instead of one per point.
__device__ __forceinline__ void chain_wrench_vjp(float *__restrict__ psum, const float *__restrict__ cumul,
                                                 const BwdTables &t, int l, f3 F, f3 T) {
  const int cs = t.chain_off[l];
  for (int ci = t.chain_off[l + 1] - 1; ci >= cs; ci--) {
    const int j = t.chain[ci];
    const int info = t.link_info[j];
    const int jt = (info & 0xff) - 1;
    if (jt < J_X_PRISM) continue;
    const float sign = t.sign[j];
    const float *C = cumul + j * 12;
    const int ax = jt >= J_X_ROT ? jt - J_X_ROT : jt;
    const f3 axis = make_f3(C[ax], C[4 + ax], C[8 + ax]);
    float r;
    if (jt >= J_X_ROT) r = sign * dot(axis, T - cross(make_f3(C[3], C[7], C[11]), F));
    else r = sign * dot(axis, F);
    atomicAdd(&psum[info >> 8], r);
  }
}

}  // namespace curobo_hip
