// dynamics_device.hpp -- inverse dynamics (body-frame RNEA) and its VJP for ONE batch element per lane, shared by
// dynamics.hip (global SoA cache) and the fused rollout kernel (the same code over LDS).
// Reference: kernels/dynamics/rnea_forward_kernel.cuh:53-292, rnea_backward_kernel.cuh:65-468, spatial_algebra.cuh,
// rnea_helpers.cuh.
#pragma once
#include "common.hpp"

namespace curobo_hip {

struct RneaArgs {
  float *tau;                 // fwd out [B, D]
  float *grad_q, *grad_qd, *grad_qdd, *grad_f_ext;  // bwd out
  const float *grad_tau;      // bwd in
  const float *q, *qd, *qdd;
  const float *fixed_transforms, *link_masses_com, *link_inertias;
  const int8_t *joint_map_type;
  const int16_t *joint_map, *link_map;
  const float *joint_offset_map, *gravity;
  const int16_t *level_links;
  float *cache;      // [L][20][B]
  float *ws_fbar, *ws_abar, *ws_vbar;  // bwd adjoints, each [L][6][B] (three bases: the fused rollout kernel places them in
                                       // different LDS regions)
  const float *f_ext;
  int batch, num_links, num_dof;
};

// per-link constants in LDS: 12 fixed transform + 4 mass/com + 6 inertia + multiplier, offset = 24 floats,
// + (joint type, joint index, parent) as ints
constexpr int kLinkFloats = 24;

struct LinkConst {
  const float *F, *mc, *in;
  float mul, off;
  int jt, ji, par;
};

__device__ __forceinline__ LinkConst link_const(const float *s_f, const int *s_i, int k) {
  LinkConst c;
  c.F = s_f + k * kLinkFloats;
  c.mc = c.F + 12;
  c.in = c.F + 16;
  c.mul = c.F[22];
  c.off = c.F[23];
  // every lane of a wavefront walks the same link: joint type, joint index and parent go to scalar registers, so that
  // what depends on them is scalar control flow (no exec-mask traffic, no per-lane selects) and the rows they address
  // have scalar base addresses
  c.jt = __builtin_amdgcn_readfirstlane(s_i[k * 3]);
  c.ji = __builtin_amdgcn_readfirstlane(s_i[k * 3 + 1]);
  c.par = __builtin_amdgcn_readfirstlane(s_i[k * 3 + 2]);
  return c;
}
__device__ __forceinline__ int order_entry(const int *order, int idx) { return __builtin_amdgcn_readfirstlane(order[idx]); }

__device__ __forceinline__ void stage_links(const RneaArgs &a, float *s_f, int *s_i) {
  const int L = a.num_links;
  for (int i = threadIdx.x; i < L * kLinkFloats; i += blockDim.x) {
    const int k = i / kLinkFloats, c = i - k * kLinkFloats;
    float v;
    if (c < 12) v = a.fixed_transforms[k * 12 + c];
    else if (c < 16) v = a.link_masses_com[k * 4 + (c - 12)];
    else if (c < 22) v = a.link_inertias[k * 8 + (c - 16)];
    else v = a.joint_offset_map[k * 2 + (c - 22)];
    s_f[i] = v;
  }
  for (int k = threadIdx.x; k < L; k += blockDim.x) {
    s_i[k * 3] = a.joint_map_type[k];
    s_i[k * 3 + 1] = a.joint_map[k];
    s_i[k * 3 + 2] = a.link_map[k];
  }
  for (int k = threadIdx.x; k < L; k += blockDim.x) s_i[L * 3 + k] = a.level_links[k];
  __syncthreads();
  // Bit 16 of an order entry: the link needs its accumulator slot in memory.  In the leaves -> root sweeps a link hands its
  // contribution to its parent in registers when the parent is the next link of the walk; only a parent with a child
  // elsewhere in the order (a branch point of the tree) receives contributions through memory.  Every other link neither
  // zeroes nor re-reads the slot (a third of the cache traffic of the forward kernel on a humanoid).
  int *order = s_i + L * 3;
  int flag[4] = {0, 0, 0, 0};  // (L <= 4 * blockDim.x: checked by the callers' LDS limits, 64-thread blocks walk <= 256 links)
  int n = 0;
  for (int idx = threadIdx.x; idx < L && n < 4; idx += blockDim.x, n++) {
    const int k = order[idx];
    for (int p = 0; p < L; p++) {
      const int c = order[p];
      if (c != k && s_i[c * 3 + 2] == k && p != idx + 1) flag[n] = 1;
    }
  }
  __syncthreads();
  n = 0;
  for (int idx = threadIdx.x; idx < L && n < 4; idx += blockDim.x, n++) order[idx] |= flag[n] << 16;
  __syncthreads();
}
__device__ __forceinline__ int order_link(int e) { return e & 0xffff; }
__device__ __forceinline__ bool order_needs_slot(int e) { return (e >> 16) != 0; }

struct Sv {  // spatial vector [angular; linear]
  f3 w, v;
};
__device__ __forceinline__ Sv sv_zero() { return Sv{make_f3(0.f, 0.f, 0.f), make_f3(0.f, 0.f, 0.f)}; }
__device__ __forceinline__ Sv operator+(Sv a, Sv b) { return Sv{a.w + b.w, a.v + b.v}; }
__device__ __forceinline__ Sv operator-(Sv a, Sv b) { return Sv{a.w - b.w, a.v - b.v}; }
__device__ __forceinline__ float sv_dot(Sv a, Sv b) { return dot(a.w, b.w) + dot(a.v, b.v); }
// (the component index is the joint's axis: the same for the whole wavefront, so these are scalar branches)
__device__ __forceinline__ float sv_get(const Sv &s, int i) {
  switch (i) {
    case 0: return s.w.x;
    case 1: return s.w.y;
    case 2: return s.w.z;
    case 3: return s.v.x;
    case 4: return s.v.y;
    default: return s.v.z;
  }
}
__device__ __forceinline__ void sv_add_at(Sv &s, int i, float x) {
  switch (i) {
    case 0: s.w.x += x; break;
    case 1: s.w.y += x; break;
    case 2: s.w.z += x; break;
    case 3: s.v.x += x; break;
    case 4: s.v.y += x; break;
    case 5: s.v.z += x; break;
    default: break;
  }
}
__device__ __forceinline__ Sv sv_unit(int i, float x) {
  Sv s = sv_zero();
  sv_add_at(s, i, x);
  return s;
}

struct Rp {  // local transform: R rotates child -> parent (row-major), p = child origin in the parent frame
  float R[9];
  f3 p;
};
template <int A1, int A2>
__device__ __forceinline__ void rotate_columns(float *R, float s, float c) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const float u = R[r * 3 + A1], w = R[r * 3 + A2];
    R[r * 3 + A1] = c * u + s * w;
    R[r * 3 + A2] = -s * u + c * w;
  }
}
// compute_local_Rp (rnea_helpers.cuh): R = R_fixed R_joint(q), p = p_fixed (+ R_fixed d for prismatic)
__device__ __forceinline__ Rp local_Rp(const float *F, int jt, float q) {
  Rp t;
  t.R[0] = F[0]; t.R[1] = F[1]; t.R[2] = F[2];
  t.R[3] = F[4]; t.R[4] = F[5]; t.R[5] = F[6];
  t.R[6] = F[8]; t.R[7] = F[9]; t.R[8] = F[10];
  t.p = make_f3(F[3], F[7], F[11]);
  if (jt >= J_X_ROT) {
    float s, c;
    sincos_bounded(q, &s, &c);
    switch (jt - J_X_ROT) {  // columns a1 = axis + 1, a2 = axis + 2 of every row: (u, w) -> (c u + s w, -s u + c w)
      case 0: rotate_columns<1, 2>(t.R, s, c); break;
      case 1: rotate_columns<2, 0>(t.R, s, c); break;
      default: rotate_columns<0, 1>(t.R, s, c); break;
    }
  } else if (jt >= J_X_PRISM) {
    switch (jt - J_X_PRISM) {
      case 0: t.p.x += F[0] * q; t.p.y += F[4] * q; t.p.z += F[8] * q; break;
      case 1: t.p.x += F[1] * q; t.p.y += F[5] * q; t.p.z += F[9] * q; break;
      default: t.p.x += F[2] * q; t.p.y += F[6] * q; t.p.z += F[10] * q; break;
    }
  }
  return t;
}
__device__ __forceinline__ f3 rot_T(const float *R, f3 v) {  // R^T v
  return make_f3(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z,
                 R[2] * v.x + R[5] * v.y + R[8] * v.z);
}
__device__ __forceinline__ f3 rot(const float *R, f3 v) {
  return make_f3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
                 R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
// X m = [E w; E (v + w x p)], E = R^T          (spatial_Xv)
__device__ __forceinline__ Sv X_motion(const Rp &t, Sv m) { return Sv{rot_T(t.R, m.w), rot_T(t.R, m.v + cross(m.w, t.p))}; }
// X^T f = [R n + p x (R f); R f]               (spatial_XTf)
__device__ __forceinline__ Sv XT_force(const Rp &t, Sv f) {
  const f3 Rf = rot(t.R, f.v);
  return Sv{rot(t.R, f.w) + cross(t.p, Rf), Rf};
}
// I m (spatial_inertia_times_vec): mc = [com, mass], in = [ixx iyy izz ixy ixz iyz] at the CoM
__device__ __forceinline__ Sv inertia_mul(const float *mc, const float *in, Sv m) {
  const f3 c = make_f3(mc[0], mc[1], mc[2]);
  const float mass = mc[3];
  const f3 h = m.v + cross(m.w, c);
  const f3 ch = cross(c, h);
  return Sv{make_f3(in[0] * m.w.x + in[3] * m.w.y + in[4] * m.w.z + mass * ch.x,
                    in[3] * m.w.x + in[1] * m.w.y + in[5] * m.w.z + mass * ch.y,
                    in[4] * m.w.x + in[5] * m.w.y + in[2] * m.w.z + mass * ch.z),
            mass * h};
}
__device__ __forceinline__ Sv crf(Sv v, Sv f) { return Sv{cross(v.w, f.w) + cross(v.v, f.v), cross(v.w, f.v)}; }  // v x* f
__device__ __forceinline__ Sv crm(Sv a, Sv b) { return Sv{cross(a.w, b.w), cross(a.v, b.w) + cross(a.w, b.v)}; }  // a x b
__device__ __forceinline__ int s_index(int jt) { return jt >= J_X_ROT ? jt - J_X_ROT : 3 + jt - J_X_PRISM; }

// SoA slots: slot(k, c)[b]
// The slot is the same for the whole wavefront (the link being walked): a row address is a scalar base plus the lane's
// 32-bit byte offset -- scalar arithmetic per row instead of a 64-bit vector multiply-add per access.
__device__ __forceinline__ const float *row_at(const float *base, size_t B, int slot, uint32_t byte_off) {
  return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base + (size_t)slot * B) + byte_off);
}
__device__ __forceinline__ float *row_at(float *base, size_t B, int slot, uint32_t byte_off) {
  return reinterpret_cast<float *>(reinterpret_cast<char *>(base + (size_t)slot * B) + byte_off);
}
__device__ __forceinline__ Sv load_sv(const float *base, size_t B, int slot, size_t b) {
  const uint32_t o = (uint32_t)b * 4u;
  return Sv{make_f3(*row_at(base, B, slot, o), *row_at(base, B, slot + 1, o), *row_at(base, B, slot + 2, o)),
            make_f3(*row_at(base, B, slot + 3, o), *row_at(base, B, slot + 4, o), *row_at(base, B, slot + 5, o))};
}
__device__ __forceinline__ void store_sv(float *base, size_t B, int slot, size_t b, Sv s) {
  const uint32_t o = (uint32_t)b * 4u;
  *row_at(base, B, slot, o) = s.w.x; *row_at(base, B, slot + 1, o) = s.w.y; *row_at(base, B, slot + 2, o) = s.w.z;
  *row_at(base, B, slot + 3, o) = s.v.x; *row_at(base, B, slot + 4, o) = s.v.y; *row_at(base, B, slot + 5, o) = s.v.z;
}

// ---- the spatial algebra of the walks as a policy.  LaneAlg: one element per lane, a spatial vector is six registers
// (everything above).  QuadAlg: one element per QUAD -- lane c of the quad (c = 0, 1, 2; lane 3 rides along) holds
// component c of the angular and of the linear part, rotations take the other components through DPP quad broadcasts,
// cross products through the two quad rotations.  A link is then ~230 instead of ~535 dependent VALU instructions and a
// batch four times the wavefronts: the walks are bound by exactly that dependent stream (30 k instructions per
// wavefront at half a wavefront per SIMD).
struct LaneAlg {
  using SvT = Sv;
  using Xf = Rp;
  __device__ __forceinline__ bool writer() const { return true; }
  __device__ __forceinline__ SvT zero() const { return sv_zero(); }
  __device__ __forceinline__ Xf xform(const float *F, int jt, float q) const { return local_Rp(F, jt, q); }
  __device__ __forceinline__ SvT motion(const Xf &t, SvT m) const { return X_motion(t, m); }
  __device__ __forceinline__ SvT force_T(const Xf &t, SvT f) const { return XT_force(t, f); }
  __device__ __forceinline__ SvT inertia(const float *mc, const float *in, SvT m) const { return inertia_mul(mc, in, m); }
  __device__ __forceinline__ SvT cross_f(SvT v, SvT f) const { return crf(v, f); }
  __device__ __forceinline__ SvT cross_m(SvT a, SvT b) const { return crm(a, b); }
  __device__ __forceinline__ float dot6(SvT a, SvT b) const { return sv_dot(a, b); }
  __device__ __forceinline__ float get(const SvT &s, int i) const { return sv_get(s, i); }
  __device__ __forceinline__ void add_at(SvT &s, int i, float x) const { sv_add_at(s, i, x); }
  __device__ __forceinline__ SvT unit(int i, float x) const { return sv_unit(i, x); }
  __device__ __forceinline__ SvT load(const float *base, size_t B, int slot, size_t b) const { return load_sv(base, B, slot, b); }
  __device__ __forceinline__ void store(float *base, size_t B, int slot, size_t b, SvT s) const { store_sv(base, B, slot, b, s); }
  __device__ __forceinline__ SvT load6(const float *p) const { return Sv{make_f3(p[0], p[1], p[2]), make_f3(p[3], p[4], p[5])}; }
  __device__ __forceinline__ void store6_neg(float *g, SvT s) const {
    g[0] = -s.w.x; g[1] = -s.w.y; g[2] = -s.w.z; g[3] = -s.v.x; g[4] = -s.v.y; g[5] = -s.v.z;
  }
};

struct SvQ {  // component c (the lane's) of [angular; linear]
  float w, v;
};
__device__ __forceinline__ SvQ operator+(SvQ a, SvQ b) { return SvQ{a.w + b.w, a.v + b.v}; }
__device__ __forceinline__ SvQ operator-(SvQ a, SvQ b) { return SvQ{a.w - b.w, a.v - b.v}; }
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// quad_perm selectors: lane i of the quad reads lane sel_i, ctrl = sel_0 | sel_1 << 2 | sel_2 << 4 | sel_3 << 6
__device__ __forceinline__ float q_bc0(float x) { return quad_perm<0x00>(x); }
__device__ __forceinline__ float q_bc1(float x) { return quad_perm<0x55>(x); }
__device__ __forceinline__ float q_bc2(float x) { return quad_perm<0xAA>(x); }
__device__ __forceinline__ float q_next(float x) { return quad_perm<0xC9>(x); }   // [1, 2, 0, 3]: component c + 1
__device__ __forceinline__ float q_next2(float x) { return quad_perm<0xD2>(x); }  // [2, 0, 1, 3]: component c + 2
__device__ __forceinline__ float q_cross(float a, float b) {  // (a x b)_c = a_{c+1} b_{c+2} - a_{c+2} b_{c+1}
  return q_next(a) * q_next2(b) - q_next2(a) * q_next(b);
}
__device__ __forceinline__ float q_sum3(float x) { return x + q_next(x) + q_next2(x); }

struct XfQ {
  float col[3];  // R[0][c], R[1][c], R[2][c]: column c (for R^T v)
  float row[3];  // R[c][0], R[c][1], R[c][2]: row c (for R v)
  float p;       // component c of the child origin in the parent frame
};

struct QuadAlg {
  int c;  // 0, 1, 2 (lane 3 of the quad computes as lane 2 and never writes)
  using SvT = SvQ;
  using Xf = XfQ;
  __device__ __forceinline__ float pick(float x, float y, float z) const { return c == 0 ? x : c == 1 ? y : z; }
  __device__ __forceinline__ bool writer() const { return c == 0 && (threadIdx.x & 3) == 0; }
  __device__ __forceinline__ SvT zero() const { return SvQ{0.f, 0.f}; }
  __device__ __forceinline__ Xf xform(const float *F, int jt, float q) const {
    const Rp t = local_Rp(F, jt, q);  // (every lane builds the whole 3 x 3: the same instructions, no exchange)
    XfQ x;
    x.col[0] = pick(t.R[0], t.R[1], t.R[2]); x.col[1] = pick(t.R[3], t.R[4], t.R[5]); x.col[2] = pick(t.R[6], t.R[7], t.R[8]);
    x.row[0] = pick(t.R[0], t.R[3], t.R[6]); x.row[1] = pick(t.R[1], t.R[4], t.R[7]); x.row[2] = pick(t.R[2], t.R[5], t.R[8]);
    x.p = pick(t.p.x, t.p.y, t.p.z);
    return x;
  }
  __device__ __forceinline__ float rot_T(const Xf &t, float v) const { return t.col[0] * q_bc0(v) + t.col[1] * q_bc1(v) + t.col[2] * q_bc2(v); }
  __device__ __forceinline__ float rot(const Xf &t, float v) const { return t.row[0] * q_bc0(v) + t.row[1] * q_bc1(v) + t.row[2] * q_bc2(v); }
  __device__ __forceinline__ SvT motion(const Xf &t, SvT m) const { return SvQ{rot_T(t, m.w), rot_T(t, m.v + q_cross(m.w, t.p))}; }
  __device__ __forceinline__ SvT force_T(const Xf &t, SvT f) const {
    const float Rf = rot(t, f.v);
    return SvQ{rot(t, f.w) + q_cross(t.p, Rf), Rf};
  }
  __device__ __forceinline__ SvT inertia(const float *mc, const float *in, SvT m) const {
    const float com = pick(mc[0], mc[1], mc[2]), mass = mc[3];
    const float h = m.v + q_cross(m.w, com);
    const float ch = q_cross(com, h);
    // row c of the inertia at the centre of mass: (ixx ixy ixz), (ixy iyy iyz), (ixz iyz izz)
    const float i0 = pick(in[0], in[3], in[4]), i1 = pick(in[3], in[1], in[5]), i2 = pick(in[4], in[5], in[2]);
    return SvQ{i0 * q_bc0(m.w) + i1 * q_bc1(m.w) + i2 * q_bc2(m.w) + mass * ch, mass * h};
  }
  __device__ __forceinline__ SvT cross_f(SvT v, SvT f) const { return SvQ{q_cross(v.w, f.w) + q_cross(v.v, f.v), q_cross(v.w, f.v)}; }
  __device__ __forceinline__ SvT cross_m(SvT a, SvT b) const { return SvQ{q_cross(a.w, b.w), q_cross(a.v, b.w) + q_cross(a.w, b.v)}; }
  __device__ __forceinline__ float dot6(SvT a, SvT b) const { return q_sum3(c < 3 ? a.w * b.w + a.v * b.v : 0.0f); }
  __device__ __forceinline__ float get(const SvT &s, int i) const {
    const float x = i < 3 ? s.w : s.v;
    const int k = i < 3 ? i : i - 3;
    return k == 0 ? q_bc0(x) : k == 1 ? q_bc1(x) : q_bc2(x);
  }
  __device__ __forceinline__ void add_at(SvT &s, int i, float x) const {
    s.w += (i == c) ? x : 0.f;
    s.v += (i == c + 3) ? x : 0.f;
  }
  __device__ __forceinline__ SvT unit(int i, float x) const {
    SvQ s{0.f, 0.f};
    add_at(s, i, x);
    return s;
  }
  __device__ __forceinline__ SvT load(const float *base, size_t B, int slot, size_t b) const {
    const uint32_t o = ((uint32_t)c * (uint32_t)B + (uint32_t)b) * 4u;  // row slot + c, element b
    return SvQ{*row_at(base, B, slot, o), *row_at(base, B, slot + 3, o)};
  }
  __device__ __forceinline__ void store(float *base, size_t B, int slot, size_t b, SvT s) const {
    if ((threadIdx.x & 3) == 3) return;
    const uint32_t o = ((uint32_t)c * (uint32_t)B + (uint32_t)b) * 4u;
    *row_at(base, B, slot, o) = s.w; *row_at(base, B, slot + 3, o) = s.v;
  }
  __device__ __forceinline__ SvT load6(const float *p) const { return SvQ{p[c], p[3 + c]}; }
  __device__ __forceinline__ void store6_neg(float *g, SvT s) const {
    if ((threadIdx.x & 3) == 3) return;
    g[c] = -s.w; g[3 + c] = -s.v;
  }
};

// Where an element's joint-space vectors live.  RneaGlobalIO: the caller's row-major [batch, dof] tensors, element b at
// b * D (a wavefront's access is a gather of 64 rows: one cache line per lane).  RneaStagedIO: copies in LDS, [joint][65]
// with the element (= lane) fastest, staged in and out by the kernel with coalesced transfers -- the walk then issues no
// uncoalesced global access at all (dynamics.hip; measured on the Unitree G1 at the C4 size: those gathers were 60 % of the
// L2 traffic of both kernels).
struct RneaGlobalIO {
  static constexpr bool kPrefetch = true;  // inputs from global memory: requested one link ahead (link_inputs)
  const RneaArgs &a;
  uint32_t o;  // byte offset of row b: b * D * 4 (the launchers check batch * dof < 2^30); the joint index is the same for
               // the whole wavefront, so an access is (scalar tensor base + joint) + this one 32-bit lane offset
  __device__ __forceinline__ RneaGlobalIO(const RneaArgs &a_, size_t b) : a(a_), o((uint32_t)b * (uint32_t)a_.num_dof * 4u) {}
  __device__ __forceinline__ const float &in(const float *t, int j) const {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t + j) + o);
  }
  __device__ __forceinline__ float &out(float *t, int j) const { return *reinterpret_cast<float *>(reinterpret_cast<char *>(t + j) + o); }
  __device__ __forceinline__ float q(int j) const { return in(a.q, j); }
  __device__ __forceinline__ float qd(int j) const { return in(a.qd, j); }
  __device__ __forceinline__ float qdd(int j) const { return in(a.qdd, j); }
  __device__ __forceinline__ float grad_tau(int j) const { return in(a.grad_tau, j); }
  __device__ __forceinline__ void tau_zero(int D) const { for (int j = 0; j < D; j++) out(a.tau, j) = 0.0f; }
  __device__ __forceinline__ void tau_add(int j, float x) const { out(a.tau, j) += x; }
  __device__ __forceinline__ void grads_zero(int D) const {
    for (int j = 0; j < D; j++) { out(a.grad_q, j) = 0.0f; out(a.grad_qd, j) = 0.0f; out(a.grad_qdd, j) = 0.0f; }
  }
  __device__ __forceinline__ void grad_q_add(int j, float x) const { out(a.grad_q, j) += x; }
  __device__ __forceinline__ void grad_qd_add(int j, float x) const { out(a.grad_qd, j) += x; }
  __device__ __forceinline__ void grad_qdd_add(int j, float x) const { out(a.grad_qdd, j) += x; }
};

// the same accessors where the "tensors" are LDS regions (the fused rollout kernel): no read-ahead
struct RneaLdsIO : RneaGlobalIO {
  static constexpr bool kPrefetch = false;
  __device__ __forceinline__ RneaLdsIO(const RneaArgs &a_, size_t b) : RneaGlobalIO(a_, b) {}
};

// RneaTransposedIO: the inputs in [dof][batch] order in global memory (a scratch the launch fills with a coalesced
// transposition): joint j of a wavefront's elements is one contiguous run -- coalesced without any LDS, which matters when the
// walks share the CUs with a kernel that lives on LDS (the C4 rollout: the self-collision kernel next to the RNEA VJP).
struct RneaTransposedIO {
  static constexpr bool kPrefetch = true;
  const float *in0, *in1, *in2;  // forward: q, qd, qdd; backward: q, qd, grad_tau -- each [dof][batch]
  size_t B;
  uint32_t ob;  // element b as a byte offset
  RneaGlobalIO out;
  __device__ __forceinline__ float at(const float *t, int j) const {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t + (size_t)j * B) + ob);
  }
  __device__ __forceinline__ float q(int j) const { return at(in0, j); }
  __device__ __forceinline__ float qd(int j) const { return at(in1, j); }
  __device__ __forceinline__ float qdd(int j) const { return at(in2, j); }
  __device__ __forceinline__ float grad_tau(int j) const { return at(in2, j); }
  __device__ __forceinline__ void tau_zero(int D) const { out.tau_zero(D); }
  __device__ __forceinline__ void tau_add(int j, float x) const { out.tau_add(j, x); }
  __device__ __forceinline__ void grads_zero(int D) const { out.grads_zero(D); }
  __device__ __forceinline__ void grad_q_add(int j, float x) const { out.grad_q_add(j, x); }
  __device__ __forceinline__ void grad_qd_add(int j, float x) const { out.grad_qd_add(j, x); }
  __device__ __forceinline__ void grad_qdd_add(int j, float x) const { out.grad_qdd_add(j, x); }
};

constexpr int kRneaStageStride = 65;  // [joint][65]: lane e of joint j sits in bank (j + e) mod 32
struct RneaStagedIO {
  static constexpr bool kPrefetch = false;  // LDS: read where they are used (carrying them a link ahead costs more than it hides)
  // INPUTS from LDS (read-only during the walk): forward in0 = q, in1 = qd, in2 = qdd; backward in0 = q, in1 = qd,
  // in2 = grad_tau.  OUTPUTS go to the caller's tensors as in RneaGlobalIO: their read-modify-writes are off the walk's
  // dependent chain, and LDS stores in the walk would alias the link constants (both LDS: the compiler then re-reads the
  // constants after every store -- measured: the forward kernel 50 % slower).
  const float *in0, *in1, *in2;
  RneaGlobalIO out;
  int e;  // element of the workgroup = lane
  __device__ __forceinline__ float q(int j) const { return in0[j * kRneaStageStride + e]; }
  __device__ __forceinline__ float qd(int j) const { return in1[j * kRneaStageStride + e]; }
  __device__ __forceinline__ float qdd(int j) const { return in2[j * kRneaStageStride + e]; }
  __device__ __forceinline__ float grad_tau(int j) const { return in2[j * kRneaStageStride + e]; }
  __device__ __forceinline__ void tau_zero(int D) const { out.tau_zero(D); }
  __device__ __forceinline__ void tau_add(int j, float x) const { out.tau_add(j, x); }
  __device__ __forceinline__ void grads_zero(int D) const { out.grads_zero(D); }
  __device__ __forceinline__ void grad_q_add(int j, float x) const { out.grad_q_add(j, x); }
  __device__ __forceinline__ void grad_qd_add(int j, float x) const { out.grad_qd_add(j, x); }
  __device__ __forceinline__ void grad_qdd_add(int j, float x) const { out.grad_qdd_add(j, x); }
};

// The joint-space inputs of the link at position idx of the walk, requested one link AHEAD of their use: they do not depend on
// the walk, and where they come from global memory (RneaTransposedIO / RneaGlobalIO) their round trip then leaves the walk's
// dependent chain.  x0, x1, x2 = the IO's three input vectors at the link's joint (0 for a link without a moving joint).
struct LinkInputs {
  float x0, x1, x2;
};
template <bool W0, bool W1, int W2, class IO>  // W2: 0 none, 1 = qdd, 2 = grad_tau (the backward IO's third vector)
__device__ __forceinline__ LinkInputs link_inputs(const IO &io, const int *s_i, const int *order, int idx) {
  const int k = order_link(order_entry(order, idx));
  const int jt = __builtin_amdgcn_readfirstlane(s_i[k * 3]), ji = __builtin_amdgcn_readfirstlane(s_i[k * 3 + 1]);
  LinkInputs in{0.0f, 0.0f, 0.0f};
  if (jt != J_FIXED && ji >= 0) {
    if (W0) in.x0 = io.q(ji);
    if (W1) in.x1 = io.qd(ji);
    if (W2 == 1) in.x2 = io.qdd(ji);
    if (W2 == 2) in.x2 = io.grad_tau(ji);
  }
  return in;
}

// One element b of a batch of B (SoA slots with element stride B): the forward sweeps.  `order` = links in level order.
// IO = where the joint-space vectors live, V = the spatial algebra (LaneAlg: the element on one lane; QuadAlg: on a quad).
template <bool HAS_FEXT, class IO, class V = LaneAlg>
__device__ __forceinline__ void rnea_forward_element_io(const RneaArgs &a, const IO &io, const float *s_f, const int *s_i,
                                                        const int *order, size_t b, size_t B, const V &alg = V());
template <bool HAS_FEXT>
__device__ __forceinline__ void rnea_forward_element(const RneaArgs &a, const float *s_f, const int *s_i, const int *order,
                                                     size_t b, size_t B) {
  rnea_forward_element_io<HAS_FEXT>(a, RneaGlobalIO(a, b), s_f, s_i, order, b, B);
}
template <bool HAS_FEXT, class IO, class V>
__device__ __forceinline__ void rnea_forward_element_io(const RneaArgs &a, const IO &io, const float *s_f, const int *s_i,
                                                        const int *order, size_t b, size_t B, const V &alg) {
  using SvT = typename V::SvT;
  const int L = a.num_links, D = a.num_dof;
  const SvT grav = alg.load6(a.gravity);
  if (alg.writer()) io.tau_zero(D);
  // sweep 1, root -> leaves: velocities and accelerations (rnea_forward_kernel.cuh:118-188)
  // The walk is a chain of dependent steps whose operands travel through the cache (HBM / L2: a round trip per link).
  // When the links come in depth-first order the parent of a link is mostly the link just processed: its state is then
  // taken from registers and the round trip leaves the dependent path (any parents-first order is correct; the
  // backends pass a depth-first one).
  int prev_k = -1;
  SvT prev_v = alg.zero(), prev_a = alg.zero();
  LinkInputs nxt{0.0f, 0.0f, 0.0f};
  if (IO::kPrefetch) nxt = link_inputs<true, true, 1>(io, s_i, order, 0);
  for (int idx = 0; idx < L; idx++) {
    const int oe = order_entry(order, idx);
    const int k = order_link(oe);
    const LinkConst c = link_const(s_f, s_i, k);
    const bool is_root = c.par < 0 || c.par == k, moving = c.jt != J_FIXED && c.ji >= 0;
    const LinkInputs in = IO::kPrefetch ? nxt : link_inputs<true, true, 1>(io, s_i, order, idx);
    if (IO::kPrefetch && idx + 1 < L) nxt = link_inputs<true, true, 1>(io, s_i, order, idx + 1);
    float qe = 0.f, qde = 0.f, qdde = 0.f;
    if (moving) {
      qe = c.mul * in.x0 + c.off;
      qde = c.mul * in.x1;
      qdde = c.mul * in.x2;
    }
    const typename V::Xf t = alg.xform(c.F, c.jt, qe);
    SvT v = alg.zero(), acc;
    if (is_root) {
      acc = alg.motion(t, grav);
    } else if (c.par == prev_k) {
      v = alg.motion(t, prev_v);
      acc = alg.motion(t, prev_a);
    } else {
      v = alg.motion(t, alg.load(a.cache, B, c.par * 20, b));
      acc = alg.motion(t, alg.load(a.cache, B, c.par * 20 + 6, b));
    }
    if (c.jt != J_FIXED) {
      const int si = s_index(c.jt);
      alg.add_at(v, si, qde);
      alg.add_at(acc, si, qdde);
      acc = acc + alg.cross_m(v, alg.unit(si, qde));  // Coriolis
    }
    alg.store(a.cache, B, k * 20, b, v);
    alg.store(a.cache, B, k * 20 + 6, b, acc);
    if (order_needs_slot(oe)) alg.store(a.cache, B, k * 20 + 12, b, alg.zero());  // children accumulate their X^T f here
    prev_k = k; prev_v = v; prev_a = acc;
  }
  // sweep 2, leaves -> root: f = I a + v x* I v (- f_ext) + children; tau = S^T f (:190-283)
  // (a link's contribution to its parent stays in registers when the parent is the next link of the walk -- in
  // reversed depth-first order it usually is -- instead of a store the parent's load would have to wait for)
  int pend_par = -1;
  SvT pend = alg.zero();
  if (IO::kPrefetch) nxt = link_inputs<true, false, 0>(io, s_i, order, L - 1);
  for (int idx = L - 1; idx >= 0; idx--) {
    const int oe = order_entry(order, idx);
    const int k = order_link(oe);
    const LinkConst c = link_const(s_f, s_i, k);
    const LinkInputs in = IO::kPrefetch ? nxt : link_inputs<true, false, 0>(io, s_i, order, idx);
    if (IO::kPrefetch && idx > 0) nxt = link_inputs<true, false, 0>(io, s_i, order, idx - 1);
    const SvT v = alg.load(a.cache, B, k * 20, b), acc = alg.load(a.cache, B, k * 20 + 6, b);
    SvT f = alg.inertia(c.mc, c.in, acc) + alg.cross_f(v, alg.inertia(c.mc, c.in, v));
    if (HAS_FEXT) f = f - alg.load6(a.f_ext + (b * L + k) * 6);
    if (order_needs_slot(oe)) f = f + alg.load(a.cache, B, k * 20 + 12, b);
    else f = f + alg.zero();  // (-0 + 0 = +0, as the stored zero gave)
    if (pend_par == k) f = f + pend;
    pend_par = -1;
    alg.store(a.cache, B, k * 20 + 12, b, f);
    const bool moving = c.jt != J_FIXED && c.ji >= 0;
    if (moving) {
      const float tj = c.mul * alg.get(f, s_index(c.jt));
      if (alg.writer()) io.tau_add(c.ji, tj);
    }
    if (!(c.par < 0 || c.par == k)) {
      const float qe = moving ? c.mul * in.x0 + c.off : 0.0f;
      const SvT up = alg.force_T(alg.xform(c.F, c.jt, qe), f);
      if (idx > 0 && order_link(order_entry(order, idx - 1)) == c.par) { pend = up; pend_par = c.par; }
      else alg.store(a.cache, B, c.par * 20 + 12, b, alg.load(a.cache, B, c.par * 20 + 12, b) + up);
    }
  }
}

// ... and the VJP.  ACCUMULATE: add to grad_q / grad_qd / grad_qdd instead of overwriting them.
template <bool HAS_FEXT, bool ACCUMULATE, class IO, class V = LaneAlg>
__device__ __forceinline__ void rnea_backward_element_io(const RneaArgs &a, const IO &io, const float *s_f, const int *s_i,
                                                         const int *order, size_t b, size_t B, const V &alg = V());
template <bool HAS_FEXT, bool ACCUMULATE = false>
__device__ __forceinline__ void rnea_backward_element(const RneaArgs &a, const float *s_f, const int *s_i, const int *order,
                                                      size_t b, size_t B) {
  rnea_backward_element_io<HAS_FEXT, ACCUMULATE>(a, RneaGlobalIO(a, b), s_f, s_i, order, b, B);
}
template <bool HAS_FEXT, bool ACCUMULATE, class IO, class V>
__device__ __forceinline__ void rnea_backward_element_io(const RneaArgs &a, const IO &io, const float *s_f, const int *s_i,
                                                         const int *order, size_t b, size_t B, const V &alg) {
  using SvT = typename V::SvT;
  const int L = a.num_links, D = a.num_dof;
  const SvT grav = alg.load6(a.gravity);
  if (!ACCUMULATE && alg.writer()) io.grads_zero(D);
  // pass 1, root -> leaves: adjoint of the force propagation (rnea_backward_kernel.cuh:151-208)
  int prev_k = -1;
  SvT prev_fb = alg.zero();
  LinkInputs nxt{0.0f, 0.0f, 0.0f};
  if (IO::kPrefetch) nxt = link_inputs<true, false, 2>(io, s_i, order, 0);
  for (int idx = 0; idx < L; idx++) {
    const int oe = order_entry(order, idx);
    const int k = order_link(oe);
    const bool slot = order_needs_slot(oe);
    const LinkConst c = link_const(s_f, s_i, k);
    const bool is_root = c.par < 0 || c.par == k, moving = c.jt != J_FIXED && c.ji >= 0;
    const LinkInputs in = IO::kPrefetch ? nxt : link_inputs<true, false, 2>(io, s_i, order, idx);
    if (IO::kPrefetch && idx + 1 < L) nxt = link_inputs<true, false, 2>(io, s_i, order, idx + 1);
    SvT fb = alg.zero();
    const int si = moving ? s_index(c.jt) : 0;
    if (moving) alg.add_at(fb, si, c.mul * in.x2);
    if (!is_root) {
      const float qe = moving ? c.mul * in.x0 + c.off : 0.0f;
      const SvT X = alg.motion(alg.xform(c.F, c.jt, qe), c.par == prev_k ? prev_fb : alg.load(a.ws_fbar, B, c.par * 6, b));
      fb = fb + X;
      if (moving) {
        const float g = c.mul * alg.dot6(X, alg.cross_f(alg.unit(si, 1.0f), alg.load(a.cache, B, k * 20 + 12, b)));
        if (alg.writer()) io.grad_q_add(c.ji, g);
      }
    }
    alg.store(a.ws_fbar, B, k * 6, b, fb);
    if (slot) {
      alg.store(a.ws_abar, B, k * 6, b, alg.zero());
      alg.store(a.ws_vbar, B, k * 6, b, alg.zero());
    }
    prev_k = k; prev_fb = fb;
    if (HAS_FEXT) alg.store6_neg(a.grad_f_ext + (b * L + k) * 6, fb);
  }
  // pass 2, leaves -> root: adjoint of the velocity / acceleration propagation (:236-465)
  int pend_par = -1;
  SvT pend_a = alg.zero(), pend_v = alg.zero();
  if (IO::kPrefetch) nxt = link_inputs<true, true, 0>(io, s_i, order, L - 1);
  for (int idx = L - 1; idx >= 0; idx--) {
    const int oe = order_entry(order, idx);
    const int k = order_link(oe);
    const bool slot = order_needs_slot(oe);
    const LinkConst c = link_const(s_f, s_i, k);
    const LinkInputs in = IO::kPrefetch ? nxt : link_inputs<true, true, 0>(io, s_i, order, idx);
    if (IO::kPrefetch && idx > 0) nxt = link_inputs<true, true, 0>(io, s_i, order, idx - 1);
    const bool is_root = c.par < 0 || c.par == k, moving = c.jt != J_FIXED && c.ji >= 0;
    const bool par_next = !is_root && idx > 0 && order_link(order_entry(order, idx - 1)) == c.par;  // this link's pushes stay in registers
    const SvT v = alg.load(a.cache, B, k * 20, b);
    const SvT fb = alg.load(a.ws_fbar, B, k * 6, b);
    SvT ab = alg.zero(), vb = alg.zero();  // (what a link without children elsewhere would read back from its slot)
    if (slot) {
      ab = alg.load(a.ws_abar, B, k * 6, b);
      vb = alg.load(a.ws_vbar, B, k * 6, b);
    }
    ab = ab + alg.inertia(c.mc, c.in, fb);
    vb = vb - alg.cross_f(fb, alg.inertia(c.mc, c.in, v)) - alg.inertia(c.mc, c.in, alg.cross_m(v, fb));
    if (pend_par == k) { ab = ab + pend_a; vb = vb + pend_v; }
    pend_par = -1;
    const int si = moving ? s_index(c.jt) : 0;
    float gq = 0.0f, gqd = 0.0f;
    if (moving) {
      const float qdk = c.mul * in.x1;
      const float gdd = c.mul * alg.get(ab, si);
      if (alg.writer()) io.grad_qdd_add(c.ji, gdd);
      gqd -= c.mul * alg.get(alg.cross_f(v, ab), si);
      vb = vb + alg.cross_f(alg.unit(si, qdk), ab);
    }
    const float qe = moving ? c.mul * in.x0 + c.off : 0.0f;
    const typename V::Xf t = alg.xform(c.F, c.jt, qe);
    const SvT S1 = alg.unit(si, 1.0f);
    if (!is_root) {
      const SvT up = alg.force_T(t, ab);
      if (par_next) pend_a = up;
      else alg.store(a.ws_abar, B, c.par * 6, b, alg.load(a.ws_abar, B, c.par * 6, b) + up);
    }
    if (moving) {  // dX/dq on the acceleration path: the parent's acceleration, or gravity at the root
      const SvT Xa = alg.motion(t, is_root ? grav : alg.load(a.cache, B, c.par * 20 + 6, b));
      gq -= c.mul * alg.dot6(ab, alg.cross_m(S1, Xa));
      gqd += c.mul * alg.get(vb, si);
    }
    if (!is_root) {
      const SvT upv = alg.force_T(t, vb);
      if (par_next) { pend_v = upv; pend_par = c.par; }
      else alg.store(a.ws_vbar, B, c.par * 6, b, alg.load(a.ws_vbar, B, c.par * 6, b) + upv);
      if (moving) gq -= c.mul * alg.dot6(vb, alg.cross_m(S1, alg.motion(t, alg.load(a.cache, B, c.par * 20, b))));
    }
    if (moving && alg.writer()) {
      io.grad_q_add(c.ji, gq);
      io.grad_qd_add(c.ji, gqd);
    }
  }
}

}  // namespace curobo_hip
