// dynamics.hip -- inverse dynamics (body-frame RNEA) and its VJP, config 4 of the baseline.
// Reference: curobo/_src/curobolib/backends/cuda_core_backend/dynamics.py:24-260,
// kernels/dynamics/rnea_forward_kernel.cuh:53-292, rnea_backward_kernel.cuh:65-468,
// spatial_algebra.cuh, rnea_helpers.cuh (and the in-tree NumPy oracle
// curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py).
//
// MI355X design.  The reference gives every batch element a slice of shared memory (12 floats per
// link forward, 30 backward) and recompiles per (num_links, num_dof, threads-per-element); a G1
// (56 links) element needs 6.7 KB of LDS backward, i.e. ~24 elements per CU.  The batch here is
// huge (points x seeds = 10^5) and each element is a strictly serial tree walk, so the kernels
// put ONE element on ONE lane and keep the per-link state (v, a, f and the adjoints) in HBM/L2 in
// a structure-of-arrays layout: element index fastest, so every access of a wavefront is one
// fully coalesced 256-byte row.  `forward_cache` is documented by the reference as an opaque
// [batch, num_links * 20] scratch handed from forward to backward; its internal layout here is
// [link][20][batch].  The adjoints live in a caller-provided workspace [link][18][batch].
// Robot constants are staged once per workgroup in LDS; any tree shape / link count works with
// one compiled kernel (links are visited in level_links order, parents before children).
#include <cstdlib>
#include "dynamics_device.hpp"

namespace curobo_hip {

template <bool HAS_FEXT>
__global__ void __launch_bounds__(256) rnea_forward_kernel(const RneaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.num_links;
  float *s_f = smem;
  int *s_i = reinterpret_cast<int *>(smem + L * kLinkFloats);
  stage_links(a, s_f, s_i);
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x, B = (size_t)a.batch;
  if (b >= B) return;
  rnea_forward_element<HAS_FEXT>(a, s_f, s_i, s_i + L * 3, b, B);
}

template <bool HAS_FEXT>
__global__ void __launch_bounds__(256) rnea_backward_kernel(const RneaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.num_links;
  float *s_f = smem;
  int *s_i = reinterpret_cast<int *>(smem + L * kLinkFloats);
  stage_links(a, s_f, s_i);
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x, B = (size_t)a.batch;
  if (b >= B) return;
  rnea_backward_element<HAS_FEXT>(a, s_f, s_i, s_i + L * 3, b, B);
}

// ---- the same walks with the element's joint-space vectors staged through LDS (RneaStagedIO): one wavefront of 64
// elements per workgroup; q / qd / qdd (grad_tau) arrive with coalesced loads (the outputs are written as before).  The walk itself is unchanged (same device functions, same arithmetic order: results are bit-identical
// to the unstaged kernels); what goes away are the per-link gathers of 64 scattered rows.
constexpr int kStagedLanes = 64;

__device__ __forceinline__ void stage_in(float *dst, const float *src, size_t b0, int n_here, int D) {
  const float *g = src + b0 * (size_t)D;
  for (int i = threadIdx.x; i < n_here * D; i += blockDim.x) {
    const int e = i / D, j = i - e * D;
    dst[j * kRneaStageStride + e] = g[i];
  }
}

// QUAD: the 64 elements of the workgroup on 64 quads (256 threads, QuadAlg: a spatial vector's components over the lanes
// of the quad) instead of on the 64 lanes of one wavefront.  Same cache / workspace layout: the two forms can be mixed.
template <bool HAS_FEXT, bool QUAD>
__global__ void __launch_bounds__(256) rnea_forward_staged_kernel(const RneaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.num_links, D = a.num_dof;
  float *s_f = smem;
  int *s_i = reinterpret_cast<int *>(smem + L * kLinkFloats);
  float *st = smem + L * (kLinkFloats + 4);
  const size_t B = (size_t)a.batch, b0 = (size_t)blockIdx.x * kStagedLanes;
  const int n_here = (int)(B - b0 < (size_t)kStagedLanes ? B - b0 : (size_t)kStagedLanes);
  const int sz = D * kRneaStageStride;
  stage_in(st, a.q, b0, n_here, D);
  stage_in(st + sz, a.qd, b0, n_here, D);
  stage_in(st + 2 * sz, a.qdd, b0, n_here, D);
  stage_links(a, s_f, s_i);  // (ends with the workgroup barrier)
  const int e = QUAD ? (int)threadIdx.x >> 2 : (int)threadIdx.x;
  if (e < n_here) {
    RneaStagedIO io{st, st + sz, st + 2 * sz, RneaGlobalIO(a, b0 + e), e};
    if (QUAD) {
      const int c = (int)threadIdx.x & 3;
      rnea_forward_element_io<HAS_FEXT>(a, io, s_f, s_i, s_i + L * 3, b0 + e, B, QuadAlg{c < 3 ? c : 2});
    } else {
      rnea_forward_element_io<HAS_FEXT>(a, io, s_f, s_i, s_i + L * 3, b0 + e, B);
    }
  }
}

template <bool HAS_FEXT, bool QUAD>
__global__ void __launch_bounds__(256) rnea_backward_staged_kernel(const RneaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.num_links, D = a.num_dof;
  float *s_f = smem;
  int *s_i = reinterpret_cast<int *>(smem + L * kLinkFloats);
  float *st = smem + L * (kLinkFloats + 4);
  const size_t B = (size_t)a.batch, b0 = (size_t)blockIdx.x * kStagedLanes;
  const int n_here = (int)(B - b0 < (size_t)kStagedLanes ? B - b0 : (size_t)kStagedLanes);
  const int sz = D * kRneaStageStride;
  stage_in(st, a.q, b0, n_here, D);
  stage_in(st + sz, a.qd, b0, n_here, D);
  stage_in(st + 2 * sz, a.grad_tau, b0, n_here, D);
  stage_links(a, s_f, s_i);
  const int e = QUAD ? (int)threadIdx.x >> 2 : (int)threadIdx.x;
  if (e < n_here) {
    RneaStagedIO io{st, st + sz, st + 2 * sz, RneaGlobalIO(a, b0 + e), e};
    if (QUAD) {
      const int c = (int)threadIdx.x & 3;
      rnea_backward_element_io<HAS_FEXT, false>(a, io, s_f, s_i, s_i + L * 3, b0 + e, B, QuadAlg{c < 3 ? c : 2});
    } else {
      rnea_backward_element_io<HAS_FEXT, false>(a, io, s_f, s_i, s_i + L * 3, b0 + e, B);
    }
  }
}

// ---- the walks over inputs transposed into a scratch (RneaTransposedIO): link constants are the only LDS
__global__ void __launch_bounds__(256) rnea_transpose_kernel(const float *in0, const float *in1, const float *in2, float *out, int B, int D,
                                                             int first) {
  // [B][D] -> [D][B] for the tensors first .. 2 (blockIdx.z), 32 x 32 tiles through LDS: coalesced on both sides
  __shared__ float tile[32][33];
  const int which = first + (int)blockIdx.z;
  const float *in = which == 0 ? in0 : which == 1 ? in1 : in2;
  float *o = out + (size_t)which * D * B;
  const int b0 = blockIdx.x * 32, j0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int b = b0 + r, j = j0 + tx;
    tile[r][tx] = (b < B && j < D) ? in[(size_t)b * D + j] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + r, b = b0 + tx;
    if (j < D && b < B) o[(size_t)j * B + b] = tile[tx][r];
  }
}

template <bool HAS_FEXT, bool QUAD, bool BACKWARD, bool ACC = false>
__global__ void __launch_bounds__(256) rnea_scratch_kernel(const RneaArgs a, const float *scratch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.num_links, D = a.num_dof;
  float *s_f = smem;
  int *s_i = reinterpret_cast<int *>(smem + L * kLinkFloats);
  stage_links(a, s_f, s_i);  // (ends with the workgroup barrier)
  const size_t B = (size_t)a.batch;
  const size_t b = (size_t)blockIdx.x * kStagedLanes + (QUAD ? threadIdx.x >> 2 : threadIdx.x);
  if (b >= B) return;
  const RneaTransposedIO io{scratch, scratch + (size_t)D * B, scratch + 2 * (size_t)D * B, B, (uint32_t)b * 4u, RneaGlobalIO(a, b)};
  if (QUAD) {
    const int c = (int)threadIdx.x & 3;
    if (BACKWARD) rnea_backward_element_io<HAS_FEXT, ACC>(a, io, s_f, s_i, s_i + L * 3, b, B, QuadAlg{c < 3 ? c : 2});
    else rnea_forward_element_io<HAS_FEXT>(a, io, s_f, s_i, s_i + L * 3, b, B, QuadAlg{c < 3 ? c : 2});
  } else {
    if (BACKWARD) rnea_backward_element_io<HAS_FEXT, ACC>(a, io, s_f, s_i, s_i + L * 3, b, B);
    else rnea_forward_element_io<HAS_FEXT>(a, io, s_f, s_i, s_i + L * 3, b, B);
  }
}

}  // namespace curobo_hip

using namespace curobo_hip;

// the scratch form of a launch: transposition, then the walk
template <bool BACKWARD>
static int launch_rnea_scratch(const RneaArgs &a, const float *in0, const float *in1, const float *in2, float *scratch, bool fext,
                               hipStream_t st, const char *what, int first = 0, bool accumulate = false) {
  const int B = a.batch, D = a.num_dof;
  hipLaunchKernelGGL(rnea_transpose_kernel, dim3((unsigned)((B + 31) / 32), (unsigned)((D + 31) / 32), (unsigned)(3 - first)), dim3(256), 0, st, in0,
                     in1, in2, scratch, B, D, first);
  // element per LANE by default here: these launches exist to run next to LDS- and wavefront-hungry kernels, and a lane walk
  // occupies a quarter of the wavefront slots of the quad walk (C4 rollout set: 900 us with lanes, 945 us with quads; alone the
  // quad walk is the faster one).  CUROBO_RNEA_SCRATCH_QUAD: 1 = quads, 2 = quads forward only, 3 = quads backward only.
  static const int quad_mode = [] { const char *e = getenv("CUROBO_RNEA_SCRATCH_QUAD"); return e ? atoi(e) : 0; }();
  const bool quad = quad_mode == 1 || (quad_mode == 2 && !BACKWARD) || (quad_mode == 3 && BACKWARD);
  const dim3 grid((unsigned)((B + kStagedLanes - 1) / kStagedLanes)), block(quad ? 4 * kStagedLanes : kStagedLanes);
  const size_t lds = (size_t)a.num_links * (kLinkFloats + 4) * sizeof(float);
  if (BACKWARD && accumulate) {  // gradients ADDED to the output tensors (the caller's running joint-space gradients)
    if (fext) { if (quad) hipLaunchKernelGGL((rnea_scratch_kernel<true, true, BACKWARD, true>), grid, block, lds, st, a, scratch);
                else hipLaunchKernelGGL((rnea_scratch_kernel<true, false, BACKWARD, true>), grid, block, lds, st, a, scratch); }
    else { if (quad) hipLaunchKernelGGL((rnea_scratch_kernel<false, true, BACKWARD, true>), grid, block, lds, st, a, scratch);
           else hipLaunchKernelGGL((rnea_scratch_kernel<false, false, BACKWARD, true>), grid, block, lds, st, a, scratch); }
    return check_launch(what, st);
  }
  if (fext) { if (quad) hipLaunchKernelGGL((rnea_scratch_kernel<true, true, BACKWARD>), grid, block, lds, st, a, scratch);
              else hipLaunchKernelGGL((rnea_scratch_kernel<true, false, BACKWARD>), grid, block, lds, st, a, scratch); }
  else { if (quad) hipLaunchKernelGGL((rnea_scratch_kernel<false, true, BACKWARD>), grid, block, lds, st, a, scratch);
         else hipLaunchKernelGGL((rnea_scratch_kernel<false, false, BACKWARD>), grid, block, lds, st, a, scratch); }
  return check_launch(what, st);
}

// LDS of the staged kernels: link constants + three joint-space input vectors [dof][65]
static size_t rnea_staged_lds(int num_links, int num_dof, int vectors) {
  return ((size_t)num_links * (kLinkFloats + 4) + (size_t)vectors * num_dof * kRneaStageStride) * sizeof(float);
}
constexpr size_t kRneaStagedLdsLimit = 80 * 1024;  // two workgroups per CU

template <class K>
static int raise_lds(K kfn, size_t lds, const char *what) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: cannot raise LDS limit: %s", what, hipGetErrorString(e));
  }
  return CUROBO_HIP_OK;
}

// elements per workgroup (= active lanes of its one wavefront below 64): EXPERIMENT knob CUROBO_RNEA_LANES
static int rnea_block(int batch_size) {
  static int forced = -1;
  if (forced < 0) { const char *e = getenv("CUROBO_RNEA_LANES"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced;
  return batch_size <= 128 * 1024 ? 64 : 256;
}

static bool rnea_staged() {  // EXPERIMENT knob: CUROBO_RNEA_STAGED=0 runs the unstaged kernels
  static int v = -1;
  if (v < 0) { const char *e = getenv("CUROBO_RNEA_STAGED"); v = e ? atoi(e) : 1; }
  return v != 0;
}

static bool rnea_quad() {  // EXPERIMENT knob: CUROBO_RNEA_QUAD=0 keeps an element on one lane
  static int v = -1;
  if (v < 0) { const char *e = getenv("CUROBO_RNEA_QUAD"); v = e ? atoi(e) : 1; }
  return v != 0;
}

static size_t rnea_lds(int num_links) { return (size_t)num_links * (kLinkFloats + 4) * sizeof(float); }

static int rnea_forward_impl(
    float *tau, const float *q, const float *qd, const float *qdd, const float *fixed_transforms,
    const float *link_masses_com, const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const float *joint_offset_map, const float *gravity, const int16_t *level_starts,
    const int16_t *level_links, float *forward_cache, int batch_size, int num_links, int num_dof, int n_levels,
    int threads_per_batch, const float *f_ext, float *scratch, curobo_hip_stream_t stream, const char *what) {
  (void)level_starts; (void)n_levels; (void)threads_per_batch;  // one lane per element: level order is all that is needed
  CUROBO_REQUIRE(num_links >= 1 && num_dof >= 1, "%s: bad dimensions", what);
  CUROBO_REQUIRE(rnea_lds(num_links) <= 64 * 1024, "%s: too many links (%d)", what, num_links);
  CUROBO_REQUIRE((long long)batch_size * (num_dof > 4 ? num_dof : 4) < (1ll << 30), "%s: batch too large for 32-bit row offsets (%d x %d)",
                 what, batch_size, num_dof);
  if (batch_size == 0) return CUROBO_HIP_OK;
  RneaArgs a{};
  a.tau = tau; a.q = q; a.qd = qd; a.qdd = qdd; a.fixed_transforms = fixed_transforms;
  a.link_masses_com = link_masses_com; a.link_inertias = link_inertias; a.joint_map_type = joint_map_type;
  a.joint_map = joint_map; a.link_map = link_map; a.joint_offset_map = joint_offset_map; a.gravity = gravity;
  a.level_links = level_links; a.cache = forward_cache; a.f_ext = f_ext;
  a.batch = batch_size; a.num_links = num_links; a.num_dof = num_dof;
  hipStream_t st = (hipStream_t)stream;
  if (scratch != nullptr) return launch_rnea_scratch<false>(a, q, qd, qdd, scratch, f_ext != nullptr, st, what);
  // one element per lane and a strictly serial walk: the launch is latency bound, so small batches are spread one
  // wavefront per workgroup over as many CUs as possible
  const size_t slds = rnea_staged_lds(num_links, num_dof, 3);
  if (slds <= kRneaStagedLdsLimit && rnea_staged()) {
    const bool quad = rnea_quad();
    const dim3 grid((unsigned)ceil_div(batch_size, kStagedLanes)), block(quad ? 4 * kStagedLanes : kStagedLanes);
#define CUROBO_RNEA_LAUNCH(FE, QD)                                                                        \
  do {                                                                                                    \
    if (int rc = raise_lds(rnea_forward_staged_kernel<FE, QD>, slds, what)) return rc;                      \
    hipLaunchKernelGGL((rnea_forward_staged_kernel<FE, QD>), grid, block, slds, st, a);                     \
  } while (0)
    if (f_ext) { if (quad) CUROBO_RNEA_LAUNCH(true, true); else CUROBO_RNEA_LAUNCH(true, false); }
    else { if (quad) CUROBO_RNEA_LAUNCH(false, true); else CUROBO_RNEA_LAUNCH(false, false); }
#undef CUROBO_RNEA_LAUNCH
    return check_launch(what, st);
  }
  const int bt = rnea_block(batch_size);
  const dim3 grid((unsigned)ceil_div(batch_size, bt)), block(bt);
  if (f_ext) hipLaunchKernelGGL((rnea_forward_kernel<true>), grid, block, rnea_lds(num_links), st, a);
  else hipLaunchKernelGGL((rnea_forward_kernel<false>), grid, block, rnea_lds(num_links), st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_rnea_forward(
    float *tau, const float *q, const float *qd, const float *qdd, const float *fixed_transforms,
    const float *link_masses_com, const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const float *joint_offset_map, const float *gravity, const int16_t *level_starts,
    const int16_t *level_links, float *forward_cache, int batch_size, int num_links, int num_dof, int n_levels,
    int threads_per_batch, const float *f_ext, curobo_hip_stream_t stream) {
  return rnea_forward_impl(tau, q, qd, qdd, fixed_transforms, link_masses_com, link_inertias, joint_map_type, joint_map, link_map,
                           joint_offset_map, gravity, level_starts, level_links, forward_cache, batch_size, num_links, num_dof, n_levels,
                           threads_per_batch, f_ext, nullptr, stream, "launch_rnea_forward");
}

CUROBO_EXPORT int curobo_hip_launch_rnea_forward_scratch(
    float *tau, const float *q, const float *qd, const float *qdd, const float *fixed_transforms,
    const float *link_masses_com, const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const float *joint_offset_map, const float *gravity, const int16_t *level_starts,
    const int16_t *level_links, float *forward_cache, int batch_size, int num_links, int num_dof, int n_levels,
    int threads_per_batch, const float *f_ext, float *scratch, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(scratch != nullptr || batch_size == 0, "launch_rnea_forward_scratch: scratch [3 * num_dof * batch_size] floats is required");
  return rnea_forward_impl(tau, q, qd, qdd, fixed_transforms, link_masses_com, link_inertias, joint_map_type, joint_map, link_map,
                           joint_offset_map, gravity, level_starts, level_links, forward_cache, batch_size, num_links, num_dof, n_levels,
                           threads_per_batch, f_ext, scratch, stream, "launch_rnea_forward_scratch");
}

static int rnea_backward_impl(
    float *grad_q, float *grad_qd, float *grad_qdd, const float *grad_tau, const float *q, const float *qd,
    const float *fixed_transforms, const float *link_masses_com, const float *link_inertias,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map, const float *joint_offset_map,
    const float *gravity, const int16_t *level_starts, const int16_t *level_links, const float *forward_cache,
    int batch_size, int num_links, int num_dof, int n_levels, int threads_per_batch, float *grad_f_ext,
    float *workspace, float *scratch, int flags, curobo_hip_stream_t stream, const char *what) {
  (void)level_starts; (void)n_levels; (void)threads_per_batch;
  CUROBO_REQUIRE(num_links >= 1 && num_dof >= 1, "%s: bad dimensions", what);
  CUROBO_REQUIRE(rnea_lds(num_links) <= 64 * 1024, "%s: too many links (%d)", what, num_links);
  CUROBO_REQUIRE((long long)batch_size * (num_dof > 4 ? num_dof : 4) < (1ll << 30), "%s: batch too large for 32-bit row offsets (%d x %d)",
                 what, batch_size, num_dof);
  CUROBO_REQUIRE(workspace != nullptr || batch_size == 0, "%s: workspace [num_links * 18 * batch_size] floats is required", what);
  if (batch_size == 0) return CUROBO_HIP_OK;
  RneaArgs a{};
  a.grad_q = grad_q; a.grad_qd = grad_qd; a.grad_qdd = grad_qdd; a.grad_f_ext = grad_f_ext; a.grad_tau = grad_tau;
  a.q = q; a.qd = qd; a.fixed_transforms = fixed_transforms; a.link_masses_com = link_masses_com;
  a.link_inertias = link_inertias; a.joint_map_type = joint_map_type; a.joint_map = joint_map; a.link_map = link_map;
  a.joint_offset_map = joint_offset_map; a.gravity = gravity; a.level_links = level_links;
  a.cache = const_cast<float *>(forward_cache);
  a.ws_fbar = workspace;  // [3][L][6][B] inside the caller's opaque [num_links * 18 * batch] workspace
  a.ws_abar = workspace + (size_t)num_links * 6 * batch_size;
  a.ws_vbar = workspace + (size_t)num_links * 12 * batch_size;
  a.batch = batch_size; a.num_links = num_links; a.num_dof = num_dof;
  hipStream_t st = (hipStream_t)stream;
  if (scratch != nullptr) return launch_rnea_scratch<true>(a, q, qd, grad_tau, scratch, grad_f_ext != nullptr, st, what, (flags & 1) ? 2 : 0, (flags & 2) != 0);
  const size_t slds = rnea_staged_lds(num_links, num_dof, 3);
  if (slds <= kRneaStagedLdsLimit && rnea_staged()) {
    const bool quad = rnea_quad();
    const dim3 grid((unsigned)ceil_div(batch_size, kStagedLanes)), block(quad ? 4 * kStagedLanes : kStagedLanes);
#define CUROBO_RNEA_LAUNCH(FE, QD)                                                                        \
  do {                                                                                                    \
    if (int rc = raise_lds(rnea_backward_staged_kernel<FE, QD>, slds, what)) return rc;                      \
    hipLaunchKernelGGL((rnea_backward_staged_kernel<FE, QD>), grid, block, slds, st, a);                     \
  } while (0)
    if (grad_f_ext) { if (quad) CUROBO_RNEA_LAUNCH(true, true); else CUROBO_RNEA_LAUNCH(true, false); }
    else { if (quad) CUROBO_RNEA_LAUNCH(false, true); else CUROBO_RNEA_LAUNCH(false, false); }
#undef CUROBO_RNEA_LAUNCH
    return check_launch(what, st);
  }
  const int bt = rnea_block(batch_size);
  const dim3 grid((unsigned)ceil_div(batch_size, bt)), block(bt);
  if (grad_f_ext) hipLaunchKernelGGL((rnea_backward_kernel<true>), grid, block, rnea_lds(num_links), st, a);
  else hipLaunchKernelGGL((rnea_backward_kernel<false>), grid, block, rnea_lds(num_links), st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_launch_rnea_backward(
    float *grad_q, float *grad_qd, float *grad_qdd, const float *grad_tau, const float *q, const float *qd,
    const float *fixed_transforms, const float *link_masses_com, const float *link_inertias,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map, const float *joint_offset_map,
    const float *gravity, const int16_t *level_starts, const int16_t *level_links, const float *forward_cache,
    int batch_size, int num_links, int num_dof, int n_levels, int threads_per_batch, float *grad_f_ext,
    float *workspace, curobo_hip_stream_t stream) {
  return rnea_backward_impl(grad_q, grad_qd, grad_qdd, grad_tau, q, qd, fixed_transforms, link_masses_com, link_inertias, joint_map_type,
                            joint_map, link_map, joint_offset_map, gravity, level_starts, level_links, forward_cache, batch_size, num_links,
                            num_dof, n_levels, threads_per_batch, grad_f_ext, workspace, nullptr, 0, stream, "launch_rnea_backward");
}

CUROBO_EXPORT int curobo_hip_launch_rnea_backward_scratch(
    float *grad_q, float *grad_qd, float *grad_qdd, const float *grad_tau, const float *q, const float *qd,
    const float *fixed_transforms, const float *link_masses_com, const float *link_inertias,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map, const float *joint_offset_map,
    const float *gravity, const int16_t *level_starts, const int16_t *level_links, const float *forward_cache,
    int batch_size, int num_links, int num_dof, int n_levels, int threads_per_batch, float *grad_f_ext,
    float *workspace, float *scratch, int flags, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(scratch != nullptr || batch_size == 0, "launch_rnea_backward_scratch: scratch [3 * num_dof * batch_size] floats is required");
  return rnea_backward_impl(grad_q, grad_qd, grad_qdd, grad_tau, q, qd, fixed_transforms, link_masses_com, link_inertias, joint_map_type,
                            joint_map, link_map, joint_offset_map, gravity, level_starts, level_links, forward_cache, batch_size, num_links,
                            num_dof, n_levels, threads_per_batch, grad_f_ext, workspace, scratch, flags, stream,
                            "launch_rnea_backward_scratch");
}
