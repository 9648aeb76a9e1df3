// seed_ik.hip -- the iteration-state update of the Levenberg-Marquardt seed-IK solver in one launch.
//
// Reference (all torch, ~25 elementwise launches per iteration under a CUDA graph):
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:338-387  joint-limit residual rows
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:292-305  pose-error reduction
//   curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:464-495  combination of the residual blocks
//   curobo/_src/solver/seed_ik/seed_iteration_state_manager.py:74-260  trust ratio, step acceptance,
//                                                                     damping update, state selection,
//                                                                     convergence flags
// One 16-lane DPP row per problem (4 problems per wavefront): lanes stride over the dofs / Jacobian
// elements, row reductions for the sums.  Everything is a pure function of the candidate buffers
// and the previous state, so the kernel is graph-capturable and deterministic.
#include <cstdlib>

#include "common.hpp"
#include "cost_device.hpp"
#include "fk_device.hpp"
#include "self_device.hpp"

namespace curobo_hip {

struct SeedIkUpdateArgs {
  // state (read-modify-write)
  float *q, *jacobian, *jTerror, *error_norm, *position_error, *orientation_error, *lambda_damping;
  uint8_t *success, *improvement;
  // candidate
  const float *cand_q, *cand_pose_jacobian, *cand_pose_jTerror, *cand_pose_cost, *cand_position_distance,
      *cand_rotation_distance, *pred_reduction;
  const float *action_min, *action_max;
  // optional velocity clamping of the limits (seed_ik_error_calculator.py:355-363)
  const float *current_position, *dt, *velocity_limits;
  // optional velocity / acceleration regularisation rows (seed_ik_error_calculator.py:389-456): with
  // v = (q - current_position) / dt,  r_v = sqrt(w_v dt) v  and  r_a = sqrt(w_a) (v - current_velocity); both blocks are
  // DIAGONAL in q, like the joint-limit block, and the LM step only sees J^T J and J^T r, so the three diagonal rows of a
  // dof are stored as ONE row of magnitude sqrt(sum of squares) (same normal equations, same predicted reduction)
  const float *current_velocity;
  float velocity_weight, acceleration_weight;
  float joint_limit_weight, rho_min, lambda_factor, lambda_min, lambda_max, conv_pos_tol, conv_ori_tol, conv_jl_weight;
  int n, D, T, initial;
};

constexpr int kRow = 16;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// velocity / acceleration residuals of dof d: squared Jacobian diagonal, J^T r and squared error
__device__ __forceinline__ void vel_acc_rows(const SeedIkUpdateArgs &a, int p, int d, float x, float &diag2, float &jtr, float &err2) {
  diag2 = jtr = err2 = 0.0f;
  if (a.current_position == nullptr || a.dt == nullptr || !(a.dt[p] > 0.0f)) return;
  if (a.velocity_weight <= 0.0f && a.acceleration_weight <= 0.0f) return;
  const float dt = fmaxf(a.dt[p], 1e-10f), inv_dt = 1.0f / dt;
  const float v = (x - a.current_position[(size_t)p * a.D + d]) * inv_dt;
  if (a.velocity_weight > 0.0f) {
    const float sw = sqrtf(a.velocity_weight * dt), e = sw * v, jd = sw * inv_dt;
    diag2 += jd * jd; jtr += jd * e; err2 += e * e;
  }
  if (a.acceleration_weight > 0.0f && a.current_velocity != nullptr) {
    const float sw = sqrtf(a.acceleration_weight), e = sw * (v - a.current_velocity[(size_t)p * a.D + d]), jd = sw * inv_dt;
    diag2 += jd * jd; jtr += jd * e; err2 += e * e;
  }
}

__device__ __forceinline__ float row16_maxf(float v) { return row16_max(v); }

// Where one problem's values live: global memory for the stand-alone launch, LDS inside the fused solver.
struct SeedIkRow {
  float *q, *J, *jTe, *error_norm, *pos_err, *ori_err, *lambda;  // state: [D], [R][D], [D], scalars
  uint8_t *success, *improvement;
  const float *cq, *cJ, *cjTe, *cpose_cost, *cpos_dist, *crot_dist, *pred;  // candidate: [D], [6T][D], [D], [2T], [T], [T], scalar
};

// The update of problem p by its 16-lane row.  Rows beyond n keep running (DPP row reductions need all lanes of a
// wave) with p clamped and live = false: no stores.
template <int DT = 0, int TT = 0>
__device__ __forceinline__ void seed_ik_update_row(const SeedIkUpdateArgs &a, const SeedIkRow &v, int p, bool live, int lane,
                                                   bool initial) {
  const int D = DT > 0 ? DT : a.D, T = TT > 0 ? TT : a.T;
  const float *cq = v.cq;

  // ---- joint-limit residual of the candidate: error, diagonal Jacobian, J^T e contribution
  float jl_sum = 0.0f;
  bool inside = true;  // strictly inside [action_min, action_max] (convergence check on the SELECTED q, below)
  for (int d = lane; d < D; d += kRow) {
    float lo = a.action_min[d], hi = a.action_max[d];
    if (a.current_position && a.velocity_limits) {
      const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
      lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
      hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
    }
    const float x = cq[d];
    const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
    jl_sum += a.joint_limit_weight * (lv + uv);
    float d2, jt, e2;
    vel_acc_rows(a, p, d, x, d2, jt, e2);
    jl_sum += e2;  // error norms of the velocity / acceleration blocks (_combine_errors, :464-495)
  }
  jl_sum = row16_sum(jl_sum);

  // ---- candidate error norm = sum of the pose cost terms + joint-limit errors; worst tool frame errors
  float pose_sum = 0.0f, pos_e = 0.0f, ori_e = 0.0f;
  for (int i = lane; i < 2 * T; i += kRow) pose_sum += v.cpose_cost[i];
  for (int t = lane; t < T; t += kRow) {
    pos_e = fmaxf(pos_e, v.cpos_dist[t]);
    ori_e = fmaxf(ori_e, v.crot_dist[t]);
  }
  pose_sum = row16_sum(pose_sum);
  pos_e = row16_maxf(pos_e);
  ori_e = row16_maxf(ori_e);
  const float cand_norm = pose_sum + jl_sum;

  // ---- trust-region ratio, acceptance, damping (seed_iteration_state_manager.py:124-180)
  bool accepted = true;
  float lambda = *v.lambda;
  if (!initial) {
    const float actual = *v.error_norm - cand_norm;
    const float rho = actual / (*v.pred + 1e-8f);
    accepted = rho >= a.rho_min;  // false for NaN
    lambda = accepted ? lambda / a.lambda_factor : lambda * a.lambda_factor;
    lambda = fminf(fmaxf(lambda, a.lambda_min), a.lambda_max);
  }

  // ---- state selection: accepted -> candidate values, rejected -> keep (error_norm is ALWAYS the
  // candidate's, seed_iteration_state_manager.py:117)
  if (accepted) {
    float *J = v.J;
    const float *cJ = v.cJ;
    if (live) {
      for (int i = lane; i < 6 * T * D; i += kRow) J[i] = cJ[i];
      for (int i = lane; i < D * D; i += kRow) J[6 * T * D + i] = 0.0f;
    }
    for (int d = lane; d < D; d += kRow) {
      float lo = a.action_min[d], hi = a.action_max[d];
      if (a.current_position && a.velocity_limits) {
        const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
        lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
        hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
      }
      const float x = cq[d];
      const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
      const float err = a.joint_limit_weight * (lv + uv);
      const float diag = a.joint_limit_weight * ((lv > 0.0f ? -1.0f : 0.0f) + (uv > 0.0f ? 1.0f : 0.0f));
      float d2, jt, e2;
      vel_acc_rows(a, p, d, x, d2, jt, e2);
      if (live) {
        v.q[d] = x;
        v.jTe[d] = v.cjTe[d] + diag * err + jt;
      }
    }
    // the zero fill above and the diagonal below touch the same elements from different lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < D; d += kRow) {
      float lo = a.action_min[d], hi = a.action_max[d];
      if (a.current_position && a.velocity_limits) {
        const float cp = a.current_position[(size_t)p * D + d], dt = a.dt[p];
        lo = fmaxf(lo, cp + a.velocity_limits[d] * dt);
        hi = fminf(hi, cp + a.velocity_limits[D + d] * dt);
      }
      const float x = cq[d];
      const float uv = fmaxf(x - hi, 0.0f), lv = fmaxf(lo - x, 0.0f);
      const float diag = a.joint_limit_weight * ((lv > 0.0f ? -1.0f : 0.0f) + (uv > 0.0f ? 1.0f : 0.0f));
      float d2, jt, e2;
      vel_acc_rows(a, p, d, x, d2, jt, e2);
      if (live) J[(size_t)(6 * T + d) * D + d] = d2 > 0.0f ? sqrtf(diag * diag + d2) : diag;
    }
  } else {
    pos_e = *v.pos_err;
    ori_e = *v.ori_err;
  }

  // ---- convergence flag on the selected state (:222-260)
  const float *sq = accepted ? cq : v.q;
  for (int d = lane; d < D; d += kRow) {
    const float x = sq[d];
    inside = inside && (x > a.action_min[d]) && (x < a.action_max[d]);
  }
  const float outside = row16_maxf(inside ? 0.0f : 1.0f);
  bool ok = pos_e < a.conv_pos_tol && ori_e < a.conv_ori_tol;
  if (a.conv_jl_weight > 0.0f) ok = ok && outside == 0.0f;
  if (live && lane == 0) {
    *v.error_norm = cand_norm;
    *v.pos_err = pos_e;
    *v.ori_err = ori_e;
    *v.lambda = lambda;
    *v.success = ok ? 1 : 0;
    *v.improvement = accepted ? 1 : 0;
  }
}


__global__ void __launch_bounds__(256) seed_ik_update_kernel(const SeedIkUpdateArgs a) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) / kRow, lane = threadIdx.x % kRow;
  const bool live = row < a.n;
  const int p = live ? row : a.n - 1;
  const int D = a.D, T = a.T, R = 6 * T + D;
  SeedIkRow v;
  v.q = a.q + (size_t)p * D; v.J = a.jacobian + (size_t)p * R * D; v.jTe = a.jTerror + (size_t)p * D;
  v.error_norm = a.error_norm + p; v.pos_err = a.position_error + p; v.ori_err = a.orientation_error + p;
  v.lambda = a.lambda_damping + p; v.success = a.success + p; v.improvement = a.improvement + p;
  v.cq = a.cand_q + (size_t)p * D; v.cJ = a.cand_pose_jacobian + (size_t)p * 6 * T * D;
  v.cjTe = a.cand_pose_jTerror + (size_t)p * D; v.cpose_cost = a.cand_pose_cost + (size_t)p * 2 * T;
  v.cpos_dist = a.cand_position_distance + (size_t)p * T; v.crot_dist = a.cand_rotation_distance + (size_t)p * T;
  v.pred = a.pred_reduction ? a.pred_reduction + p : nullptr;
  seed_ik_update_row(a, v, p, live, lane, a.initial != 0);
}


// ------------------------------------------------------------------------------------------------
// The whole Levenberg-Marquardt iteration -- and any number of them -- in ONE launch.
//
// As launches an iteration is five dependent kernels on ~10^4 small problems (LM step, FK + Jacobian, tool-pose
// error, J^T e through the chain, state update): ~9 us each of which almost all is launch and memory latency,
// 47 us per iteration, 0.75 ms for the reference's 16 iterations.  Nothing of an iteration is shared between
// problems, so here a problem lives on one 16-lane row (four per wavefront) with its whole state in LDS -- the
// accepted Jacobian [6T + D][D], the normal matrix, q, J^T e, the 13 link transforms of the candidate -- and
// the row runs `iterations` iterations back to back; global memory sees the state once on the way in and once
// on the way out.  Same arithmetic per stage as the stand-alone kernels (shared device functions:
// fk_chain_16, tool_pose_distance_point, seed_ik_update_row); J^T J is contracted on the matrix cores here too
// (a wavefront takes the Jacobians of its four rows in turn), the D x D system is then solved in registers.
struct SeedIkSolveArgs {
  SeedIkUpdateArgs u;  // state pointers and parameters (u.cand_q = the seeds when u.initial; other u.cand_* unused)
  ToolPoseArgs tp;     // goal set, weights (current_* / out_* unused)
  const float *fixed_transform, *joint_offset;
  const int8_t *joint_map_type;
  const int16_t *joint_map, *link_map, *tool_frame_map, *link_chain_data, *link_chain_offsets, *joint_links_data,
      *joint_links_offsets;
  const uint8_t *joint_affects_endeffector;
  int L, iterations, chain_len;
  // optional device-side early exit: the launch returns at once when *stop_flag != 0, else counts itself in *blocks_run
  const int32_t *stop_flag;
  int32_t *blocks_run;
};

__host__ __device__ inline int seed_ik_row_floats(int D, int T, int L) {
  const int R = 6 * T + D;
  // J | A [D][D+1] | q | q_cand | jTe | cand jTe | cand J | cumul | locals | pose (2T + T + T + 3T + 4T) | scalars + flags (8) | bcast (16)
  return R * D + D * (D + 1) + 4 * D + 6 * T * D + L * 12 + L * 16 + 11 * T + 8 + 16;
}

// dynamic LDS of one 256-thread workgroup of seed_ik_solve_kernel: the shared robot tables + 16 rows of per-problem state
constexpr int kSeedIkLdsLimit = 64 * 1024;
__host__ inline size_t seed_ik_iterate_lds(int D, int T, int L, int chain_len) {
  const int rows = 256 / kRow;
  return ((size_t)((4 * L + 1 + chain_len + 3) & ~3) + (size_t)rows * ((seed_ik_row_floats(D, T, L) + 3) & ~3)) * sizeof(float);
}

#define SEED_ROW_SYNC()                                         \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
  } while (0)

// DT, TT > 0: dof / tool-frame count at compile time (every loop over them unrolls, the normal equations are solved in
// registers); 0: run-time sizes
template <int DT, int TT>
__global__ void __launch_bounds__(256, 4) seed_ik_solve_kernel(const SeedIkSolveArgs a) {  // 4 waves / SIMD: all 800 workgroups of a
                                                                                         // 100 x 128 batch resident at once
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SeedIkUpdateArgs &u = a.u;
  const int D = DT > 0 ? DT : u.D, T = TT > 0 ? TT : u.T, L = a.L, R = 6 * T + D, LD = D + 1;
  if (a.stop_flag != nullptr && *a.stop_flag != 0) return;  // the batch was solved by an earlier block of iterations
  const int tid = threadIdx.x, rowi = tid / kRow, lane = tid % kRow;
  if (a.blocks_run != nullptr && blockIdx.x == 0 && tid == 0) atomicAdd(a.blocks_run, 1);
  const int row = blockIdx.x * (blockDim.x / kRow) + rowi;
  const bool live = row < u.n;
  const int p = live ? row : u.n - 1;
  // robot tables shared by the rows: parent, (joint type + 1) | joint << 8, axis sign, chain offsets, chain links
  int *s_parent = reinterpret_cast<int *>(smem);
  int *s_info = s_parent + L;
  float *s_sign = reinterpret_cast<float *>(s_info + L);
  int *s_choff = reinterpret_cast<int *>(s_sign + L);
  int *s_chain = s_choff + (L + 1);
  float *base = smem + ((4 * L + 1 + a.chain_len + 3) & ~3) + (size_t)rowi * ((seed_ik_row_floats(D, T, L) + 3) & ~3);
  float *sJ = base, *sA = sJ + R * D, *sq = sA + D * LD, *sqc = sq + D, *sjte = sqc + D, *scjte = sjte + D;
  float *scJ = scjte + D, *cumul = scJ + 6 * T * D, *locals = cumul + L * 12, *pose = locals + L * 16;
  float *pose_cost = pose, *pos_dist = pose + 2 * T, *rot_dist = pos_dist + T, *gpos = rot_dist + T, *gquat = gpos + 3 * T;
  float *scal = gquat + 4 * T;  // [0] error_norm [1] pos_err [2] ori_err [3] lambda [4] pred; flags behind
  uint8_t *flags = reinterpret_cast<uint8_t *>(scal + 6);
  float *bc = scal + 8;  // [16] row broadcast scratch
  for (int l = tid; l < L; l += blockDim.x) {
    s_parent[l] = a.link_map[l];
    s_info[l] = ((int)a.joint_map_type[l] + 1) | ((int)(a.joint_map[l] < 0 ? 0 : a.joint_map[l]) << 8);
    s_sign[l] = a.joint_offset[2 * l];
  }
  for (int l = tid; l <= L; l += blockDim.x) s_choff[l] = a.link_chain_offsets[l];
  for (int i = tid; i < a.chain_len; i += blockDim.x) s_chain[i] = a.link_chain_data[i];
  // ---- state in
  if (!u.initial) {
    for (int i = lane; i < R * D; i += kRow) sJ[i] = u.jacobian[(size_t)p * R * D + i];
    for (int d = lane; d < D; d += kRow) { sq[d] = u.q[(size_t)p * D + d]; sjte[d] = u.jTerror[(size_t)p * D + d]; }
    if (lane == 0) { scal[0] = u.error_norm[p]; scal[1] = u.position_error[p]; scal[2] = u.orientation_error[p]; }
  }
  if (lane == 0) { scal[3] = u.lambda_damping[p]; scal[4] = 0.0f; flags[0] = 0; flags[1] = 0; }
  __syncthreads();  // (s_parent; the rows are independent from here on)
  SeedIkRow v;
  v.q = sq; v.J = sJ; v.jTe = sjte; v.error_norm = scal; v.pos_err = scal + 1; v.ori_err = scal + 2; v.lambda = scal + 3;
  v.success = flags; v.improvement = flags + 1;
  v.cq = sqc; v.cJ = scJ; v.cjTe = scjte; v.cpose_cost = pose_cost; v.cpos_dist = pos_dist; v.crot_dist = rot_dist; v.pred = scal + 4;

  const int n_eval = a.iterations + (u.initial ? 1 : 0);
  for (int it = 0; it < n_eval; it++) {
    const bool initial = u.initial && it == 0;
    if (initial) {
      for (int d = lane; d < D; d += kRow) sqc[d] = u.cand_q[(size_t)p * D + d];
    } else {
      // ---- LM step: A = J^T J + lambda I, Cholesky, two triangular solves (levenberg_marquardt_step.py:146-199)
      const float lam = scal[3];
      // J^T J on the matrix cores, as in the stand-alone LM step (linalg.hip): the wavefront contracts the Jacobians of
      // its four rows one after the other with v_mfma_f32_16x16x4_f32 (A and B operand are the same register: both are
      // 4 rows of J), lane (kk, col) feeding J[k0 + kk][col]; the D x D corner of the 16 x 16 result goes to that
      // row's normal matrix.  Same accumulation order as the stand-alone kernel: identical J^T J.
      {
        const int lane64 = tid & 63, kk = lane64 >> 4, col = lane64 & 15;
        const size_t row_stride = (size_t)((seed_ik_row_floats(D, T, L) + 3) & ~3);
        float *wave_base = base - (size_t)(rowi & 3) * row_stride;  // row 0 of this wavefront
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const float *Jr = wave_base + rr * row_stride;
          float *Ar = wave_base + rr * row_stride + R * D;
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
          for (int k0 = 0; k0 < R; k0 += 4) {
            const int r = k0 + kk;
            const float x = (r < R && col < D) ? Jr[r * D + col] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc, 0, 0, 0);
          }
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {  // C layout: column = lane & 15, row = (lane >> 4) * 4 + reg
            const int ri = kk * 4 + reg;
            if (ri < D && col < D) Ar[ri * LD + col] = acc[reg];
          }
        }
      }
      SEED_ROW_SYNC();
      const int i = lane;
      float dlt = 0.0f, g = 0.0f;
      if (DT > 0) {
        // every lane factorises the D x D system by itself in registers: no cross-lane step, no barrier (as a row-
        // distributed factorisation it is 2 D barriers for the Cholesky and 4 D for the solves, each an LDS round trip)
        constexpr int DD = DT > 0 ? DT : 1;
        float Lm[DD][DD], gv[DD], yv[DD];
#pragma unroll
        for (int r = 0; r < DD; r++) {
          gv[r] = sjte[r];
#pragma unroll
          for (int cidx = 0; cidx <= r; cidx++) Lm[r][cidx] = sA[r * LD + cidx];
        }
#pragma unroll
        for (int j = 0; j < DD; j++) {
          float sv = Lm[j][j] + lam;
#pragma unroll
          for (int k = 0; k < j; k++) sv -= Lm[j][k] * Lm[j][k];
          const float ljj = sqrtf(sv);
          Lm[j][j] = ljj;
#pragma unroll
          for (int r = j + 1; r < DD; r++) {
            float sr = Lm[r][j];
#pragma unroll
            for (int k = 0; k < j; k++) sr -= Lm[r][k] * Lm[j][k];
            Lm[r][j] = sr / ljj;
          }
        }
#pragma unroll
        for (int j = 0; j < DD; j++) {  // L y = -g
          float y = -gv[j];
#pragma unroll
          for (int k = 0; k < j; k++) y -= Lm[j][k] * yv[k];
          yv[j] = y / Lm[j][j];
        }
#pragma unroll
        for (int j = DD - 1; j >= 0; j--) {  // L^T delta = y
          float dv = yv[j];
#pragma unroll
          for (int k = j + 1; k < DD; k++) dv -= Lm[k][j] * yv[k];
          yv[j] = dv / Lm[j][j];
        }
#pragma unroll
        for (int j = 0; j < DD; j++) {
          if (i == j) { dlt = yv[j]; g = gv[j]; }
        }
      } else {
        for (int j = 0; j < D; j++) {  // lane i = row i (left-looking), two row barriers per column
          float sv = 0.0f;
          if (i >= j && i < D) {
            sv = sA[i * LD + j] + (i == j ? lam : 0.0f);
            for (int k = 0; k < j; k++) sv -= sA[i * LD + k] * sA[j * LD + k];
          }
          if (i == j) sA[j * LD + j] = sqrtf(sv);
          SEED_ROW_SYNC();
          if (i > j && i < D) sA[i * LD + j] = sv / sA[j * LD + j];
          SEED_ROW_SYNC();
        }
        g = i < D ? sjte[i] : 0.0f;
        float y = -g;
        for (int j = 0; j < D; j++) {
          if (i == j) bc[0] = y / sA[j * LD + j];
          SEED_ROW_SYNC();
          const float yj = bc[0];
          if (i == j) y = yj;
          if (i > j && i < D) y -= sA[i * LD + j] * yj;
          SEED_ROW_SYNC();
        }
        dlt = y;
        for (int j = D - 1; j >= 0; j--) {
          if (i == j) bc[0] = dlt / sA[j * LD + j];
          SEED_ROW_SYNC();
          const float dj = bc[0];
          if (i == j) dlt = dj;
          if (i < j) dlt -= sA[j * LD + i] * dj;
          SEED_ROW_SYNC();
        }
      }
      if (i < D) sqc[i] = sq[i] + dlt;
      const float red = row16_sum(i < D ? dlt * (lam * dlt - g) : 0.0f);
      if (lane == 0) scal[4] = 0.5f * red;
    }
    SEED_ROW_SYNC();
    // ---- FK of the candidate on the row (kinematics_forward_helper.cuh:316-512)
    for (int l = lane; l < L; l += kRow) {
      const int info = s_info[l];
      const int jt = (info & 0xff) - 1;
      const float qv = jt != J_FIXED ? sqc[info >> 8] : 0.0f;
      local_transform_colmajor(locals + l * 16, a.fixed_transform + l * 12, jt, qv, s_sign[l], a.joint_offset[2 * l + 1]);
    }
    for (int d = lane; d < D; d += kRow) scjte[d] = 0.0f;
    SEED_ROW_SYNC();
    {
      float *const cm[1] = {cumul};
      const float *const lc[1] = {locals};
      fk_chain_16_multi<1>(cm, lc, s_parent, a.fixed_transform, L, lane);  // (four links' operands per LDS round trip)
    }
    SEED_ROW_SYNC();
    // ---- tool-frame Jacobian columns (kinematics_forward_kernel.cuh:45-200): column j = sum over the links of joint j
    // that lie on the tool's chain (link 0 excluded), so a lane takes a chain entry and adds to its joint's column
    for (int i = lane; i < 6 * T * D; i += kRow) scJ[i] = 0.0f;
    SEED_ROW_SYNC();
    for (int t = 0; t < T; t++) {
      const int tl = a.tool_frame_map[t];
      const float *E = cumul + tl * 12;
      const f3 ee = make_f3(E[3], E[7], E[11]);
      const int cs = s_choff[tl], ce = s_choff[tl + 1];
      for (int ci = cs + lane; ci < ce; ci += kRow) {
        const int li = s_chain[ci];
        const int info = s_info[li];
        const int jt = (info & 0xff) - 1;
        if (li == 0 || jt < J_X_PRISM) continue;
        const int j = info >> 8;
        const float *C = cumul + li * 12;
        const float sign = s_sign[li];
        float *col = scJ + (t * 6) * D + j;
        if (jt >= J_X_ROT) {
          const int ax = jt - J_X_ROT;
          const f3 axis = sign * make_f3(C[ax], C[4 + ax], C[8 + ax]);
          const f3 lin = cross(axis, ee - make_f3(C[3], C[7], C[11]));
          atomicAdd(col, lin.x); atomicAdd(col + D, lin.y); atomicAdd(col + 2 * D, lin.z);
          atomicAdd(col + 3 * D, axis.x); atomicAdd(col + 4 * D, axis.y); atomicAdd(col + 5 * D, axis.z);
        } else {
          const int ax = jt - J_X_PRISM;
          atomicAdd(col, sign * C[ax]); atomicAdd(col + D, sign * C[4 + ax]); atomicAdd(col + 2 * D, sign * C[8 + ax]);
        }
      }
    }
    // ---- tool-pose error (wp_tool_pose.py:456-692), one tool frame per lane
    for (int t = lane; t < T; t += kRow) {
      const float *C = cumul + a.tool_frame_map[t] * 12;
      const float4 qx = quat_from_transform(C);
      const ToolPoseResult r = tool_pose_distance_point(a.tp, p, 0, t, make_f3(C[3], C[7], C[11]), make_float4(qx.w, qx.x, qx.y, qx.z));
      pose_cost[2 * t] = r.position_cost; pose_cost[2 * t + 1] = r.rotation_cost;
      pos_dist[t] = r.position_distance; rot_dist[t] = r.rotation_distance;
      gpos[3 * t] = r.position_gradient.x; gpos[3 * t + 1] = r.position_gradient.y; gpos[3 * t + 2] = r.position_gradient.z;
      gquat[4 * t] = r.quat_rate_wxyz.x; gquat[4 * t + 1] = r.quat_rate_wxyz.y; gquat[4 * t + 2] = r.quat_rate_wxyz.z;
      gquat[4 * t + 3] = r.quat_rate_wxyz.w;
    }
    SEED_ROW_SYNC();
    // ---- J^T e of the pose error through the chain (kinematics_backward_helper.cuh:102-183)
    for (int t = 0; t < T; t++) {
      const f3 g = make_f3(gpos[3 * t], gpos[3 * t + 1], gpos[3 * t + 2]);
      const float dqw = gquat[4 * t], dqx = gquat[4 * t + 1], dqy = gquat[4 * t + 2], dqz = gquat[4 * t + 3];
      if (g.x == 0.f && g.y == 0.f && g.z == 0.f && dqw == 0.f && dqx == 0.f && dqy == 0.f && dqz == 0.f) continue;
      const int l = a.tool_frame_map[t];
      const float *C = cumul + l * 12;
      const float4 qx = quat_from_transform(C);
      const f3 pos = make_f3(C[3], C[7], C[11]);
      const f3 om = make_f3(0.5f * (-qx.x * dqw + qx.w * dqx + qx.z * dqy - qx.y * dqz),
                            0.5f * (-qx.y * dqw - qx.z * dqx + qx.w * dqy + qx.x * dqz),
                            0.5f * (-qx.z * dqw + qx.y * dqx - qx.x * dqy + qx.w * dqz));
      const int cs = s_choff[l], ce = s_choff[l + 1];
      for (int ci = cs + lane; ci < ce; ci += kRow) {
        const int j = s_chain[ci];
        const int info = s_info[j];
        const int jt = (info & 0xff) - 1;
        if (jt < J_X_PRISM) continue;
        const float sign = s_sign[j];
        const float *Cj = cumul + j * 12;
        const int ax = jt >= J_X_ROT ? jt - J_X_ROT : jt;
        const f3 axis = make_f3(Cj[ax], Cj[4 + ax], Cj[8 + ax]);
        float r;
        if (jt >= J_X_ROT) r = dot(sign * g, cross(axis, pos - make_f3(Cj[3], Cj[7], Cj[11]))) + sign * dot(axis, om);
        else r = sign * dot(axis, g);
        atomicAdd(&scjte[info >> 8], r);
      }
    }
    SEED_ROW_SYNC();
    // ---- trust ratio, acceptance, damping, state selection, convergence (seed_iteration_state_manager.py:74-260)
    seed_ik_update_row<DT, TT>(u, v, p, true, lane, initial);
    SEED_ROW_SYNC();
  }
  // ---- state out
  if (live) {
    for (int i = lane; i < R * D; i += kRow) u.jacobian[(size_t)p * R * D + i] = sJ[i];
    for (int d = lane; d < D; d += kRow) { u.q[(size_t)p * D + d] = sq[d]; u.jTerror[(size_t)p * D + d] = sjte[d]; }
    if (lane == 0) {
      u.error_norm[p] = scal[0]; u.position_error[p] = scal[1]; u.orientation_error[p] = scal[2];
      u.lambda_damping[p] = scal[3]; u.success[p] = flags[0]; u.improvement[p] = flags[1];
    }
  }
}


// reference _calculate_exit_condition (seed_ik_solver.py:452-468) on the device: *stop_flag = 1 when at least
// `needed` problems have a converged seed.  One workgroup; the next curobo_hip_seed_ik_iterate launches return at
// once when the flag is set, so the host enqueues every block of iterations without waiting for this answer.
__global__ void __launch_bounds__(256) seed_ik_batch_status_kernel(const uint8_t *success, int P, int S, int needed, int32_t *stop_flag) {
  __shared__ int s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  // one 16-lane row per problem and round, lanes stride over its seeds (coalesced byte loads), row-level OR
  const int rowi = threadIdx.x / kRow, lane = threadIdx.x % kRow, rows = blockDim.x / kRow;
  int mine = 0;
  for (int p0 = 0; p0 < P; p0 += rows) {
    const int p = p0 + rowi;
    float any = 0.0f;
    if (p < P)
      for (int k = lane; k < S; k += kRow) any = fmaxf(any, success[(size_t)p * S + k] != 0 ? 1.0f : 0.0f);
    any = row16_max(any);
    if (lane == 0 && any > 0.0f) mine++;
  }
  if (mine) atomicAdd(&s_count, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_count >= needed) *stop_flag = 1;
}


// ------------------------------------------------------------------------------------------------
// Ranking of the seeds of a problem in ONE launch (the reference does it with ~20 torch kernels: comparisons,
// masks, a top-k, gathers).  One workgroup per problem: every seed's cost goes to LDS, a thread finds the rank of
// its seed by counting the seeds that beat it (cost, then lower index: the order of a stable sort, independent of
// anything but the values), and the k best are written out in rank order with what the caller wants of them.
constexpr int kRankMaxSeeds = 1024;

__device__ __forceinline__ int rank_of(const float *s_cost, int S, int i) {
  const float ci = s_cost[i];
  int r = 0;
  for (int j = 0; j < S; j++) {
    const float cj = s_cost[j];
    r += (cj < ci || (cj == ci && j < i)) ? 1 : 0;
  }
  return r;
}

struct SeedSelectArgs {
  uint8_t *out_success;
  float *out_solution, *out_pos, *out_ori;
  const float *q, *pos_err, *ori_err, *lim_lo, *lim_hi, *current_position;
  float pos_tol, ori_tol, cspace_w;
  int P, S, D, k, check_limits;
};

// reference SeedIKSolver._select_top_solutions (seed_ik_solver.py:522-572)
__global__ void __launch_bounds__(256) seed_ik_select_kernel(const SeedSelectArgs a) {
  __shared__ float s_cost[kRankMaxSeeds];
  __shared__ uint8_t s_ok[kRankMaxSeeds];
  const int p = blockIdx.x, S = a.S, D = a.D;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const size_t e = (size_t)p * S + i;
    const float pe = a.pos_err[e], oe = a.ori_err[e];
    bool ok = pe < a.pos_tol && oe < a.ori_tol;
    float dist2 = 0.0f;
    for (int d = 0; d < D; d++) {
      const float x = a.q[e * D + d];
      if (a.check_limits) ok = ok && x > a.lim_lo[d] && x < a.lim_hi[d];
      if (a.current_position) { const float df = x - a.current_position[(size_t)p * D + d]; dist2 += df * df; }
    }
    float c = pe + oe;
    if (a.current_position && a.cspace_w > 0.0f) c = c + a.cspace_w * sqrtf(dist2);
    c = c + 1e10f * (ok ? 0.0f : 1.0f);
    s_cost[i] = c == c ? c : __builtin_inff();
    s_ok[i] = ok ? 1 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const int r = rank_of(s_cost, S, i);
    if (r < a.k) {
      const size_t e = (size_t)p * S + i, o = (size_t)p * a.k + r;
      a.out_success[o] = s_ok[i];
      a.out_pos[o] = a.pos_err[e];
      a.out_ori[o] = a.ori_err[e];
      for (int d = 0; d < D; d++) a.out_solution[o * D + d] = a.q[e * D + d];
    }
  }
}

struct IkRankArgs {
  uint8_t *out_success;
  float *out_solution, *out_pos, *out_rot, *out_cost;
  int64_t *out_seed, *out_goalset;
  const float *q, *cost, *pos_dist, *rot_dist, *self_dist, *cspace_cost, *scene_dist;
  const int32_t *goalset_idx;
  float pos_thr, rot_thr;
  int P, S, D, T, n_scene, k;
  int64_t seed_offset;
};

// reference IKSolver._get_result ranking (solver_ik.py:440-580): feasible = no self collision, no joint-limit
// cost, no scene collision; success = feasible and within the pose thresholds; ranked by cost + 1e16 (not success)
__global__ void __launch_bounds__(256) ik_rank_kernel(const IkRankArgs a) {
  __shared__ float s_cost[kRankMaxSeeds];
  __shared__ uint8_t s_ok[kRankMaxSeeds];
  const int p = blockIdx.x, S = a.S, D = a.D, T = a.T;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const size_t e = (size_t)p * S + i;
    float cs = 0.0f;
    for (int d = 0; d < D; d++) cs += a.cspace_cost[e * D + d];
    bool ok = a.self_dist[e] <= 0.0f && cs <= 0.0f;
    if (a.scene_dist) {
      float sc = 0.0f;
      for (int j = 0; j < a.n_scene; j++) sc += a.scene_dist[e * a.n_scene + j];
      ok = ok && sc <= 0.0f;
    }
    // converged on EVERY tool frame (reference: torch.all over the per-link convergence list, solver_ik.py:463-476)
    for (int t = 0; t < T; t++) ok = ok && a.pos_dist[e * T + t] < a.pos_thr && a.rot_dist[e * T + t] < a.rot_thr;
    const float c = a.cost[e] + 1e16f * (ok ? 0.0f : 1.0f);
    s_cost[i] = c == c ? c : __builtin_inff();
    s_ok[i] = ok ? 1 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const int r = rank_of(s_cost, S, i);
    if (r < a.k) {
      const size_t e = (size_t)p * S + i, o = (size_t)p * a.k + r;
      a.out_success[o] = s_ok[i];
      float pe = a.pos_dist[e * T], re = a.rot_dist[e * T];  // the largest error over the tool frames (solver_ik.py:549-550)
      for (int t = 1; t < T; t++) { pe = fmaxf(pe, a.pos_dist[e * T + t]); re = fmaxf(re, a.rot_dist[e * T + t]); }
      a.out_pos[o] = pe;
      a.out_rot[o] = re;
      a.out_cost[o] = a.cost[e];
      a.out_seed[o] = a.seed_offset + i;
      for (int t = 0; t < T; t++) a.out_goalset[o * T + t] = a.goalset_idx ? (int64_t)a.goalset_idx[e * T + t] : 0;
      for (int d = 0; d < D; d++) a.out_solution[o * D + d] = a.q[e * D + d];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Arg-min over the seeds of a problem with its payload row, one launch: row[p] = (min cost, seed_offset + first index
// that attains it, payload[p, index, :]).  The local stage of the seed-parallel arg-min exchange (reference single-GPU
// equivalent: solver_ik.py:503-515, solver_trajopt.py:469-484); torch needs ~8 small kernels for it.
__global__ void __launch_bounds__(256) argmin_rows_kernel(float *row, const float *cost, const float *payload, int S, int V,
                                                          float seed_offset) {
  __shared__ float s_c[4];
  __shared__ int s_i[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  float bc = __builtin_inff();
  int bi = 0x7fffffff;
  for (int i = tid; i < S; i += blockDim.x) {
    const float c = cost[(size_t)p * S + i];
    if (c < bc || (c == bc && i < bi)) { bc = c; bi = i; }  // (NaN never wins; all-NaN rows fall through to index 0 below)
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const float oc = __shfl_xor(bc, off, kWave);
    const int oi = __shfl_xor(bi, off, kWave);
    if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
  }
  if ((tid & (kWave - 1)) == 0) { s_c[tid / kWave] = bc; s_i[tid / kWave] = bi; }
  __syncthreads();
  bc = s_c[0]; bi = s_i[0];
  for (int w = 1; w < (int)(blockDim.x / kWave); w++)
    if (s_c[w] < bc || (s_c[w] == bc && s_i[w] < bi)) { bc = s_c[w]; bi = s_i[w]; }
  if (bi == 0x7fffffff) { bi = 0; bc = cost[(size_t)p * S]; }
  float *out = row + (size_t)p * (2 + V);
  if (tid == 0) { out[0] = bc; out[1] = seed_offset + (float)bi; }
  for (int v = tid; v < V; v += blockDim.x) out[2 + v] = payload[((size_t)p * S + bi) * V + v];
}

}  // namespace curobo_hip

using namespace curobo_hip;

CUROBO_EXPORT int curobo_hip_seed_ik_update_state(
    float *joint_position, float *jacobian, float *jTerror, float *error_norm, float *position_error,
    float *orientation_error, float *lambda_damping, uint8_t *success, uint8_t *improvement,
    const float *candidate_joint_position, const float *candidate_pose_jacobian, const float *candidate_pose_jTerror,
    const float *candidate_pose_cost, const float *candidate_position_distance, const float *candidate_rotation_distance,
    const float *predicted_reduction, const float *action_min, const float *action_max, const float *current_position,
    const float *dt, const float *velocity_limits, const float *current_velocity, float velocity_weight,
    float acceleration_weight, float joint_limit_weight, float rho_min, float lambda_factor,
    float lambda_min, float lambda_max, float convergence_position_tolerance, float convergence_orientation_tolerance,
    float convergence_joint_limit_weight, int num_problems, int dof, int num_tool_frames, int initial,
    curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(num_problems >= 0 && dof >= 1 && num_tool_frames >= 1, "seed_ik_update_state: bad sizes (n=%d, dof=%d, T=%d)",
                 num_problems, dof, num_tool_frames);
  CUROBO_REQUIRE(initial || predicted_reduction, "seed_ik_update_state: predicted_reduction is NULL%s", "");
  CUROBO_REQUIRE(!current_position || dt, "seed_ik_update_state: current_position needs dt%s", "");
  if (num_problems == 0) return CUROBO_HIP_OK;
  SeedIkUpdateArgs a;
  a.q = joint_position; a.jacobian = jacobian; a.jTerror = jTerror; a.error_norm = error_norm;
  a.position_error = position_error; a.orientation_error = orientation_error; a.lambda_damping = lambda_damping;
  a.success = success; a.improvement = improvement;
  a.cand_q = candidate_joint_position; a.cand_pose_jacobian = candidate_pose_jacobian; a.cand_pose_jTerror = candidate_pose_jTerror;
  a.cand_pose_cost = candidate_pose_cost; a.cand_position_distance = candidate_position_distance;
  a.cand_rotation_distance = candidate_rotation_distance; a.pred_reduction = predicted_reduction;
  a.action_min = action_min; a.action_max = action_max; a.current_position = current_position; a.dt = dt;
  a.velocity_limits = velocity_limits;
  a.current_velocity = current_velocity; a.velocity_weight = velocity_weight; a.acceleration_weight = acceleration_weight;
  a.joint_limit_weight = joint_limit_weight; a.rho_min = rho_min; a.lambda_factor = lambda_factor; a.lambda_min = lambda_min;
  a.lambda_max = lambda_max; a.conv_pos_tol = convergence_position_tolerance; a.conv_ori_tol = convergence_orientation_tolerance;
  a.conv_jl_weight = convergence_joint_limit_weight;
  a.n = num_problems; a.D = dof; a.T = num_tool_frames; a.initial = initial;
  hipStream_t st = (hipStream_t)stream;
  const int rows_per_block = 256 / kRow;
  hipLaunchKernelGGL(seed_ik_update_kernel, dim3(ceil_div(num_problems, rows_per_block)), dim3(256), 0, st, a);
  return check_launch("seed_ik_update_state", st);
}

CUROBO_EXPORT int curobo_hip_seed_ik_iterate(
    float *joint_position, float *jacobian, float *jTerror, float *error_norm, float *position_error,
    float *orientation_error, float *lambda_damping, uint8_t *success, uint8_t *improvement, const float *seed_joint_position,
    const float *goal_position, const float *goal_quat, const int32_t *idxs_goal, const float *position_orientation_weight,
    const float *pose_axes_weight_factor, const float *pose_convergence_tolerance, const uint8_t *project_distance_to_goal,
    int num_goalset, int rotation_method, const float *fixed_transform, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_chain_data, const int16_t *link_chain_offsets,
    const int16_t *joint_links_data, const int16_t *joint_links_offsets, const uint8_t *joint_affects_endeffector,
    const float *joint_offset_map, const float *action_min, const float *action_max, const float *current_position,
    const float *dt, const float *velocity_limits, const float *current_velocity, float velocity_weight,
    float acceleration_weight, float joint_limit_weight, float rho_min, float lambda_factor, float lambda_min,
    float lambda_max, float convergence_position_tolerance, float convergence_orientation_tolerance,
    float convergence_joint_limit_weight, int num_problems, int dof, int num_links, int num_tool_frames, int link_chain_len,
    int iterations, int initial, const int32_t *stop_flag, int32_t *blocks_run, curobo_hip_stream_t stream) {
  const char *what = "seed_ik_iterate";
  CUROBO_REQUIRE(num_problems >= 0 && dof >= 1 && dof <= kRow && num_tool_frames >= 1 && num_links >= 1,
                 "%s: bad sizes (n=%d, dof=%d (<= 16), T=%d, L=%d)", what, num_problems, dof, num_tool_frames, num_links);
  CUROBO_REQUIRE(iterations >= 0 && (iterations > 0 || initial), "%s: nothing to do (iterations=%d, initial=%d)", what, iterations, initial);
  CUROBO_REQUIRE(!initial || seed_joint_position, "%s: initial evaluation needs seed_joint_position", what);
  CUROBO_REQUIRE(!current_position || dt, "%s: current_position needs dt", what);
  CUROBO_REQUIRE(rotation_method >= 0 && rotation_method <= 2 && num_goalset >= 1, "%s: bad goal set / rotation method", what);
  if (num_problems == 0) return CUROBO_HIP_OK;
  SeedIkSolveArgs a{};
  SeedIkUpdateArgs &u = a.u;
  u.q = joint_position; u.jacobian = jacobian; u.jTerror = jTerror; u.error_norm = error_norm;
  u.position_error = position_error; u.orientation_error = orientation_error; u.lambda_damping = lambda_damping;
  u.success = success; u.improvement = improvement; u.cand_q = seed_joint_position;
  u.action_min = action_min; u.action_max = action_max; u.current_position = current_position; u.dt = dt;
  u.velocity_limits = velocity_limits; u.current_velocity = current_velocity; u.velocity_weight = velocity_weight;
  u.acceleration_weight = acceleration_weight; u.joint_limit_weight = joint_limit_weight; u.rho_min = rho_min;
  u.lambda_factor = lambda_factor; u.lambda_min = lambda_min; u.lambda_max = lambda_max;
  u.conv_pos_tol = convergence_position_tolerance; u.conv_ori_tol = convergence_orientation_tolerance;
  u.conv_jl_weight = convergence_joint_limit_weight; u.n = num_problems; u.D = dof; u.T = num_tool_frames; u.initial = initial;
  ToolPoseArgs &tp = a.tp;
  tp.goal_position = goal_position; tp.goal_quat = goal_quat; tp.idxs_goal = idxs_goal;
  tp.position_orientation_weight = position_orientation_weight; tp.terminal_axes_weight = pose_axes_weight_factor;
  tp.non_terminal_axes_weight = pose_axes_weight_factor; tp.terminal_tolerance = pose_convergence_tolerance;
  tp.non_terminal_tolerance = pose_convergence_tolerance; tp.project_distance_to_goal = project_distance_to_goal;
  tp.batch = num_problems; tp.horizon = 1; tp.num_links = num_tool_frames; tp.num_goalset = num_goalset;
  tp.rotation_method = rotation_method;
  a.fixed_transform = fixed_transform; a.joint_offset = joint_offset_map; a.joint_map_type = joint_map_type;
  a.joint_map = joint_map; a.link_map = link_map; a.tool_frame_map = tool_frame_map; a.link_chain_data = link_chain_data;
  a.link_chain_offsets = link_chain_offsets; a.joint_links_data = joint_links_data; a.joint_links_offsets = joint_links_offsets;
  a.joint_affects_endeffector = joint_affects_endeffector; a.L = num_links; a.iterations = iterations;
  a.chain_len = link_chain_len; a.stop_flag = stop_flag; a.blocks_run = blocks_run;
  const int rows = 256 / kRow;
  const size_t lds = seed_ik_iterate_lds(dof, num_tool_frames, num_links, link_chain_len);
  CUROBO_REQUIRE(lds <= (size_t)kSeedIkLdsLimit, "%s: the per-problem state does not fit in LDS (%zu bytes); use the launch sequence", what, lds);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ceil_div(num_problems, rows)), block(256);
  if (dof == 7 && num_tool_frames == 1) hipLaunchKernelGGL((seed_ik_solve_kernel<7, 1>), grid, block, lds, st, a);
  else if (dof == 6 && num_tool_frames == 1) hipLaunchKernelGGL((seed_ik_solve_kernel<6, 1>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((seed_ik_solve_kernel<0, 0>), grid, block, lds, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_seed_ik_iterate_fits(int dof, int num_links, int num_tool_frames, int link_chain_len) {
  if (dof < 1 || dof > 16 || num_links < 1 || num_tool_frames < 1 || link_chain_len < 0) return 0;
  return seed_ik_iterate_lds(dof, num_tool_frames, num_links, link_chain_len) <= (size_t)kSeedIkLdsLimit ? 1 : 0;
}

CUROBO_EXPORT int curobo_hip_seed_ik_batch_status(const uint8_t *success, int num_problems, int num_seeds, int needed,
                                                  int32_t *stop_flag, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(success && stop_flag && num_problems >= 1 && num_seeds >= 1, "seed_ik_batch_status: bad arguments (P=%d, S=%d)",
                 num_problems, num_seeds);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seed_ik_batch_status_kernel, dim3(1), dim3(256), 0, st, success, num_problems, num_seeds, needed, stop_flag);
  return check_launch("seed_ik_batch_status", st);
}

CUROBO_EXPORT int curobo_hip_seed_ik_select(
    uint8_t *out_success, float *out_solution, float *out_position_error, float *out_orientation_error,
    const float *joint_position, const float *position_error, const float *orientation_error, const float *limit_lower,
    const float *limit_upper, const float *current_position, float position_tolerance, float orientation_tolerance,
    float start_cspace_dist_weight, int check_limits, int num_problems, int num_seeds, int dof, int return_seeds,
    curobo_hip_stream_t stream) {
  const char *what = "seed_ik_select";
  CUROBO_REQUIRE(num_problems >= 0 && num_seeds >= 1 && num_seeds <= kRankMaxSeeds && dof >= 1 && return_seeds >= 1 &&
                     return_seeds <= num_seeds, "%s: bad sizes (P=%d, S=%d (<= 1024), D=%d, k=%d)", what, num_problems, num_seeds, dof, return_seeds);
  CUROBO_REQUIRE(!check_limits || (limit_lower && limit_upper), "%s: check_limits needs the limits", what);
  if (num_problems == 0) return CUROBO_HIP_OK;
  SeedSelectArgs a{out_success, out_solution, out_position_error, out_orientation_error, joint_position, position_error,
                   orientation_error, limit_lower, limit_upper, current_position, position_tolerance, orientation_tolerance,
                   start_cspace_dist_weight, num_problems, num_seeds, dof, return_seeds, check_limits};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seed_ik_select_kernel, dim3(num_problems), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_ik_rank(
    uint8_t *out_success, float *out_solution, float *out_position_error, float *out_rotation_error, float *out_cost,
    int64_t *out_seed_index, int64_t *out_goalset_index, const float *joint_position, const float *cost,
    const float *position_distance, const float *rotation_distance, const float *self_collision_distance,
    const float *cspace_cost, const float *scene_distance, const int32_t *goalset_idx, float position_threshold,
    float rotation_threshold, int num_problems, int num_seeds, int dof, int num_tool_frames, int num_scene_columns,
    int return_seeds, int seed_offset, curobo_hip_stream_t stream) {
  const char *what = "ik_rank";
  CUROBO_REQUIRE(num_problems >= 0 && num_seeds >= 1 && num_seeds <= kRankMaxSeeds && dof >= 1 && num_tool_frames >= 1 &&
                     return_seeds >= 1 && return_seeds <= num_seeds,
                 "%s: bad sizes (P=%d, S=%d (<= 1024), D=%d, T=%d, k=%d)", what, num_problems, num_seeds, dof, num_tool_frames, return_seeds);
  if (num_problems == 0) return CUROBO_HIP_OK;
  IkRankArgs a{out_success, out_solution, out_position_error, out_rotation_error, out_cost, out_seed_index, out_goalset_index,
               joint_position, cost, position_distance, rotation_distance, self_collision_distance, cspace_cost,
               scene_distance, goalset_idx, position_threshold, rotation_threshold, num_problems, num_seeds, dof,
               num_tool_frames, num_scene_columns, return_seeds, seed_offset};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ik_rank_kernel, dim3(num_problems), dim3(256), 0, st, a);
  return check_launch(what, st);
}

CUROBO_EXPORT int curobo_hip_argmin_rows(float *out_rows, const float *cost, const float *payload, int num_problems, int num_seeds,
                                         int payload_width, int seed_offset, curobo_hip_stream_t stream) {
  CUROBO_REQUIRE(out_rows && cost && (payload || payload_width == 0) && num_problems >= 0 && num_seeds >= 1 && payload_width >= 0,
                 "argmin_rows: bad arguments (P=%d, S=%d, V=%d)", num_problems, num_seeds, payload_width);
  if (num_problems == 0) return CUROBO_HIP_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(argmin_rows_kernel, dim3(num_problems), dim3(256), 0, st, out_rows, cost, payload, num_seeds, payload_width,
                     (float)seed_offset);
  return check_launch("argmin_rows", st);
}
