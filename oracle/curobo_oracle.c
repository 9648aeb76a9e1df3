/*
 * curobo_oracle.c -- CPU restatement (plain C99, fp32) of the cuRobo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * curobo_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product path (curobo_amd) never links, imports or calls it.
 *
 * Every function cites the reference file:line (relative to NVlabs/curobo v0.8.0,
 * /root/reference) whose arithmetic it follows.  The reference kernels are CUDA / Warp; both are
 * run here FROM THEIR OWN SOURCES to pin this file:
 *   - the CUDA kernels (FK, FK VJP, self collision, B-spline) on the CPU through the CUDA-on-CPU shim
 *     of oracle/cuda_on_cpu -> oracle/_ref/libcurobo_ref.so: bit-identical outputs but for the
 *     tree-ordered sums of the VJP / centre of mass (tests/test_reference_cuda_kernels.py),
 *   - the Warp kernels (scene collision, sweep, speed metric, voxel lookup, tool pose, c-space, LM
 *     step) through the Warp stand-in of tests/golden/warp_emulator (tests/test_reference_warp_*.py),
 *   - the reference's own importable torch / NumPy twins for L-BFGS, the Wolfe line searches, MPPI,
 *     RNEA and seed IK (tests/golden/*.npz), its FK known-answer vector, and central finite
 *     differences for every VJP (the reference's own test style).
 *
 * Index tie rule (canonical): when several self-collision pairs share the
 * maximum penetration value the pair with the LOWEST index in pair_locations wins.  The
 * reference's reduction order is launch dependent (collision_pair.cuh:55-57), so the oracle and
 * the HIP kernels both implement this deterministic rule.
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math; OpenMP over the point axis for the
 * cpu_baseline timing only -- results do not depend on the thread count).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* joint types: curobolib/kernels/kinematics/kinematics_constants.h:10-16 */
#define J_FIXED (-1)
#define J_X_PRISM 0
#define J_Y_PRISM 1
#define J_Z_PRISM 2
#define J_X_ROT 3
#define J_Y_ROT 4
#define J_Z_ROT 5

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * A1. forward kinematics
 * ---------------------------------------------------------------------------------------- */

/* kinematics_forward_helper.cuh:316-393 (compute_local_link_transform) and
 * kinematics_util.cuh:62-74 (update_axis_direction).  M is row-major 3x4. */
static void orc_local_transform(float *M, const float *q_point, const float *F, int j_type,
                                int j_idx, const float *offset2) {
  const float f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3];
  const float f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7];
  const float f8 = F[8], f9 = F[9], f10 = F[10], f11 = F[11];
  if (j_type == J_FIXED) {
    memcpy(M, F, 12 * sizeof(float));
    return;
  }
  float angle = q_point[j_idx];
  angle = offset2[0] * angle + offset2[1];
  if (j_type <= J_Z_PRISM) {
    M[0] = f0; M[1] = f1; M[2] = f2;
    M[4] = f4; M[5] = f5; M[6] = f6;
    M[8] = f8; M[9] = f9; M[10] = f10;
    M[3] = f3 + (j_type == J_X_PRISM ? f0 : (j_type == J_Y_PRISM ? f1 : f2)) * angle;
    M[7] = f7 + (j_type == J_X_PRISM ? f4 : (j_type == J_Y_PRISM ? f5 : f6)) * angle;
    M[11] = f11 + (j_type == J_X_PRISM ? f8 : (j_type == J_Y_PRISM ? f9 : f10)) * angle;
    return;
  }
  const float s = sinf(angle), c = cosf(angle);
  const int xyz = j_type - J_X_ROT;
  const float is_x = xyz == 0 ? 1.0f : 0.0f;
  const float is_y = xyz == 1 ? 1.0f : 0.0f;
  const float is_z = xyz == 2 ? 1.0f : 0.0f;
  const float c0 = is_x + c * (is_y + is_z);
  const float c1 = is_y + c * (is_x + is_z);
  const float c2 = is_z + c * (is_x + is_y);
  /* column 0 */
  M[0] = f0 * c0 + s * (is_z * f1 - is_y * f2);
  M[4] = f4 * c0 + s * (is_z * f5 - is_y * f6);
  M[8] = f8 * c0 + s * (is_z * f9 - is_y * f10);
  /* column 1 */
  M[1] = f1 * c1 + s * (is_x * f2 - is_z * f0);
  M[5] = f5 * c1 + s * (is_x * f6 - is_z * f4);
  M[9] = f9 * c1 + s * (is_x * f10 - is_z * f8);
  /* column 2 */
  M[2] = f2 * c2 + s * (is_y * f0 - is_x * f1);
  M[6] = f6 * c2 + s * (is_y * f4 - is_x * f5);
  M[10] = f10 * c2 + s * (is_y * f8 - is_x * f9);
  M[3] = f3; M[7] = f7; M[11] = f11;
}

/* kinematics_forward_helper.cuh:437-465 (compose_link_transform_halfwarp):
 * child[r][c] = dot4(parent_row_r, (M[0][c], M[1][c], M[2][c], c==3)) */
static void orc_compose(float *child, const float *parent, const float *M) {
  for (int r = 0; r < 3; r++) {
    const float *p = parent + r * 4;
    for (int c = 0; c < 4; c++) {
      const float w = (c == 3) ? 1.0f : 0.0f;
      child[r * 4 + c] = p[0] * M[c] + p[1] * M[4 + c] + p[2] * M[8 + c] + p[3] * w;
    }
  }
}

/* common/quaternion_util.cuh:110-160 + :50-57 (normalise, sign so that w >= 0).
 * R is the rotation block of a row-major 3x4 (stride 4).  out is xyzw. */
static void orc_quat_from_transform(const float *T, float *qxyzw) {
  const float t0 = T[0], t1 = T[1], t2 = T[2];
  const float t3 = T[4], t4 = T[5], t5 = T[6];
  const float t6 = T[8], t7 = T[9], t8 = T[10];
  float x, y, z, w, n, ns;
  if (t8 < 0.0f) {
    if (t0 > t4) {
      n = 1 + t0 - t4 - t8;
      ns = 0.5f / sqrtf(n);
      x = n * ns; y = (t1 + t3) * ns; z = (t6 + t2) * ns; w = -1 * (t5 - t7) * ns;
    } else {
      n = 1 - t0 + t4 - t8;
      ns = 0.5f / sqrtf(n);
      x = (t1 + t3) * ns; y = n * ns; z = (t5 + t7) * ns; w = -1 * (t6 - t2) * ns;
    }
  } else {
    if (t0 < -1 * t4) {
      n = 1 - t0 - t4 + t8;
      ns = 0.5f / sqrtf(n);
      x = (t6 + t2) * ns; y = (t5 + t7) * ns; z = n * ns; w = -1 * (t1 - t3) * ns;
    } else {
      n = 1 + t0 + t4 + t8;
      ns = 0.5f / sqrtf(n);
      x = (t5 - t7) * ns; y = (t6 - t2) * ns; z = (t1 - t3) * ns; w = -1 * n * ns;
    }
  }
  float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
  if (w < 0.0f) inv = -inv;
  qxyzw[0] = x * inv; qxyzw[1] = y * inv; qxyzw[2] = z * inv; qxyzw[3] = w * inv;
}

/* kinematics_util.cuh:38-50 (transform_sphere_float4) */
static void orc_transform_sphere(const float *C, const float *sph, float *out) {
  out[0] = C[0] * sph[0] + C[1] * sph[1] + C[2] * sph[2] + C[3];
  out[1] = C[4] * sph[0] + C[5] * sph[1] + C[6] * sph[2] + C[7];
  out[2] = C[8] * sph[0] + C[9] * sph[1] + C[10] * sph[2] + C[11];
  out[3] = sph[3];
}

/*
 * orc_kinematics_forward: restates kinematics_forward_spheres_jacobian_kernel and its two
 * reduced variants (kinematics_forward_kernel.cuh:20-433).
 *   link_pos[N,T,3], link_quat[N,T,4] (wxyz), spheres_out[N,S,4], com_out[N,4],
 *   jacobian_out[N,T,6,D], cumul_out[N,L,3,4]; any output pointer may be NULL.
 *   n_points N = batch * horizon.  robot_spheres[num_envs,S,4]; env_query_idx[batch].
 */
ORC_API void orc_kinematics_forward(
    float *link_pos, float *link_quat, float *spheres_out, float *com_out, float *jacobian_out,
    float *cumul_out, const float *q, const float *fixed_transform, const float *robot_spheres,
    const float *link_masses_com, const int8_t *joint_map_type, const int16_t *joint_map,
    const int16_t *link_map, const int16_t *tool_frame_map, const int16_t *link_sphere_map,
    const int16_t *link_chain_data, const int16_t *link_chain_offsets,
    const int16_t *joint_links_data, const int16_t *joint_links_offsets,
    const uint8_t *joint_affects_endeffector, const float *joint_offset,
    const int32_t *env_query_idx, int n_points, int horizon, int nspheres, int num_envs,
    int nlinks, int njoints, int n_tool_frames) {
#pragma omp parallel
  {
    float *cumul = (float *)malloc((size_t)nlinks * 12 * sizeof(float));
#pragma omp for schedule(static)
    for (int n = 0; n < n_points; n++) {
      float M[12];
      /* base: kinematics_forward_helper.cuh:467-485 */
      memcpy(cumul, fixed_transform, 12 * sizeof(float));
      /* chain in index order: :487-512 */
      for (int l = 1; l < nlinks; l++) {
        orc_local_transform(M, q + (size_t)n * njoints, fixed_transform + l * 12,
                            joint_map_type[l], joint_map[l], joint_offset + 2 * l);
        orc_compose(cumul + l * 12, cumul + link_map[l] * 12, M);
      }
      if (cumul_out) memcpy(cumul_out + (size_t)n * nlinks * 12, cumul, nlinks * 12 * sizeof(float));
      /* spheres: kinematics_forward_helper.cuh:218-254 */
      if (spheres_out && nspheres > 0) {
        const int env = (num_envs > 1) ? env_query_idx[n / horizon] : 0;
        const float *rs = robot_spheres + (size_t)env * nspheres * 4;
        for (int s = 0; s < nspheres; s++) {
          orc_transform_sphere(cumul + link_sphere_map[s] * 12, rs + s * 4,
                               spheres_out + ((size_t)n * nspheres + s) * 4);
        }
      }
      /* CoM: kinematics_forward_helper.cuh:538-600 */
      if (com_out) {
        float acc[4] = {0, 0, 0, 0};
        for (int l = 0; l < nlinks; l++) {
          const float *mc = link_masses_com + l * 4;
          const float mass = mc[3];
          if (mass > 0.0f) {
            float cw[4];
            orc_transform_sphere(cumul + l * 12, mc, cw);
            acc[0] += mass * cw[0]; acc[1] += mass * cw[1]; acc[2] += mass * cw[2];
            acc[3] += mass;
          }
        }
        float *o = com_out + (size_t)n * 4;
        if (acc[3] > 0.0f) {
          o[0] = acc[0] / acc[3]; o[1] = acc[1] / acc[3]; o[2] = acc[2] / acc[3]; o[3] = acc[3];
        } else {
          o[0] = o[1] = o[2] = o[3] = 0.0f;
        }
      }
      /* tool poses: kinematics_forward_helper.cuh:270-301 */
      for (int t = 0; t < n_tool_frames; t++) {
        const float *C = cumul + tool_frame_map[t] * 12;
        if (link_quat) {
          float qx[4];
          orc_quat_from_transform(C, qx);
          float *o = link_quat + ((size_t)n * n_tool_frames + t) * 4;
          o[0] = qx[3]; o[1] = qx[0]; o[2] = qx[1]; o[3] = qx[2];
        }
        if (link_pos) {
          float *o = link_pos + ((size_t)n * n_tool_frames + t) * 3;
          o[0] = C[3]; o[1] = C[7]; o[2] = C[11];
        }
      }
      /* geometric Jacobian: kinematics_forward_helper.cuh:45-200 */
      if (jacobian_out) {
        for (int t = 0; t < n_tool_frames; t++) {
          const int tl = tool_frame_map[t];
          const float *E = cumul + tl * 12;
          const float ex = E[3], ey = E[7], ez = E[11];
          const int cs = link_chain_offsets[tl], ce = link_chain_offsets[tl + 1];
          float *J = jacobian_out + ((size_t)n * n_tool_frames + t) * 6 * njoints;
          for (int j = 0; j < njoints; j++) {
            float col[6] = {0, 0, 0, 0, 0, 0};
            if (joint_affects_endeffector[j * n_tool_frames + t]) {
              for (int jl = joint_links_offsets[j]; jl < joint_links_offsets[j + 1]; jl++) {
                const int li = joint_links_data[jl];
                if (li == 0) continue;
                int in_chain = 0;
                for (int ci = cs; ci < ce; ci++)
                  if (link_chain_data[ci] == li) { in_chain = 1; break; }
                if (!in_chain) continue;
                const float *C = cumul + li * 12;
                const int jt = joint_map_type[li];
                const float sign = joint_offset[li * 2];
                if (jt >= J_X_ROT && jt <= J_Z_ROT) {
                  const int a = jt - J_X_ROT;
                  const float ax = sign * C[a], ay = sign * C[4 + a], az = sign * C[8 + a];
                  const float dx = ex - C[3], dy = ey - C[7], dz = ez - C[11];
                  col[0] += ay * dz - az * dy;
                  col[1] += az * dx - ax * dz;
                  col[2] += ax * dy - ay * dx;
                  col[3] += ax; col[4] += ay; col[5] += az;
                } else if (jt >= J_X_PRISM && jt <= J_Z_PRISM) {
                  const int a = jt - J_X_PRISM;
                  col[0] += sign * C[a]; col[1] += sign * C[4 + a]; col[2] += sign * C[8 + a];
                }
              }
            }
            for (int r = 0; r < 6; r++) J[r * njoints + j] = col[r];
          }
        }
      }
    }
    free(cumul);
  }
}

/* ------------------------------------------------------------------------------------------
 * A2. forward-kinematics VJP (kinematics_backward_kernel.cuh:27-157,
 *     kinematics_backward_helper.cuh:14-291, kinematics_joint_util.cuh:13-66)
 *     grad_jacobian (dJ/dq) is not restated: "next" row in SURVEY.md section 8f-2.
 * ---------------------------------------------------------------------------------------- */
static float orc_rot_bwd_translation(const float *C, int xyz, const float *p, const float *g,
                                     float sign) {
  /* kinematics_joint_util.cuh:13-39: dot(sign*g, cross(axis, p - origin)) */
  const float vx = C[xyz], vy = C[4 + xyz], vz = C[8 + xyz];
  const float jx = p[0] - C[3], jy = p[1] - C[7], jz = p[2] - C[11];
  const float cx = vy * jz - vz * jy, cy = vz * jx - vx * jz, cz = vx * jy - vy * jx;
  return (sign * g[0]) * cx + (sign * g[1]) * cy + (sign * g[2]) * cz;
}

static float orc_prism_bwd(const float *C, int xyz, const float *g, float sign) {
  /* kinematics_joint_util.cuh:57-66 */
  return sign * (C[xyz] * g[0] + C[4 + xyz] * g[1] + C[8 + xyz] * g[2]);
}

ORC_API void orc_kinematics_backward(
    float *grad_q, const float *grad_link_pos, const float *grad_link_quat,
    const float *grad_spheres, const float *grad_com, const float *batch_com,
    const float *cumul_in, const float *robot_spheres, const float *link_masses_com,
    const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *tool_frame_map,
    const int16_t *link_sphere_map, const int16_t *link_chain_data,
    const int16_t *link_chain_offsets, const float *joint_offset, const int32_t *env_query_idx,
    int n_points, int horizon, int nspheres, int num_envs, int nlinks, int njoints,
    int n_tool_frames, int compute_com) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < n_points; n++) {
    float psum[256];
    for (int j = 0; j < njoints; j++) psum[j] = 0.0f;
    const float *cumul = cumul_in + (size_t)n * nlinks * 12;
    /* spheres: kinematics_backward_helper.cuh:14-99 */
    if (grad_spheres && nspheres > 0) {
      const int env = (num_envs > 1) ? env_query_idx[n / horizon] : 0;
      const float *rs = robot_spheres + (size_t)env * nspheres * 4;
      for (int s = 0; s < nspheres; s++) {
        const float *g = grad_spheres + ((size_t)n * nspheres + s) * 4;
        if (g[0] == 0 && g[1] == 0 && g[2] == 0) continue;
        const int l = link_sphere_map[s];
        float p[4];
        orc_transform_sphere(cumul + l * 12, rs + s * 4, p);
        for (int ci = link_chain_offsets[l + 1] - 1; ci >= link_chain_offsets[l]; ci--) {
          const int j = link_chain_data[ci];
          const int jt = joint_map_type[j];
          const float sign = joint_offset[j * 2];
          if (jt >= J_X_ROT && jt <= J_Z_ROT)
            psum[joint_map[j]] += orc_rot_bwd_translation(cumul + j * 12, jt - J_X_ROT, p, g, sign);
          else if (jt >= J_X_PRISM && jt <= J_Z_PRISM)
            psum[joint_map[j]] += orc_prism_bwd(cumul + j * 12, jt, g, sign);
        }
      }
    }
    /* tool frames: kinematics_backward_helper.cuh:102-183 */
    for (int t = 0; t < n_tool_frames; t++) {
      const float *gp = grad_link_pos + ((size_t)n * n_tool_frames + t) * 3;
      const float *gq = grad_link_quat + ((size_t)n * n_tool_frames + t) * 4;
      if (gp[0] == 0 && gp[1] == 0 && gp[2] == 0 && gq[0] == 0 && gq[1] == 0 && gq[2] == 0 &&
          gq[3] == 0)
        continue;
      const int l = tool_frame_map[t];
      const float *C = cumul + l * 12;
      float qx[4];
      orc_quat_from_transform(C, qx);
      const float pos[3] = {C[3], C[7], C[11]};
      /* quaternion_util.cuh:86-102: omega = 0.5 * E(q)^T g (q xyzw, g wxyz) */
      const float dqw = gq[0], dqx = gq[1], dqy = gq[2], dqz = gq[3];
      float om[3];
      om[0] = 0.5f * (-qx[0] * dqw + qx[3] * dqx + qx[2] * dqy - qx[1] * dqz);
      om[1] = 0.5f * (-qx[1] * dqw - qx[2] * dqx + qx[3] * dqy + qx[0] * dqz);
      om[2] = 0.5f * (-qx[2] * dqw + qx[1] * dqx - qx[0] * dqy + qx[3] * dqz);
      for (int ci = link_chain_offsets[l]; ci < link_chain_offsets[l + 1]; ci++) {
        const int j = link_chain_data[ci];
        const int jt = joint_map_type[j];
        const float sign = joint_offset[j * 2];
        const float *Cj = cumul + j * 12;
        if (jt >= J_X_ROT && jt <= J_Z_ROT) {
          const int a = jt - J_X_ROT;
          float r = orc_rot_bwd_translation(Cj, a, pos, gp, sign);
          r += sign * (Cj[a] * om[0] + Cj[4 + a] * om[1] + Cj[8 + a] * om[2]);
          psum[joint_map[j]] += r;
        } else if (jt >= J_X_PRISM && jt <= J_Z_PRISM) {
          psum[joint_map[j]] += orc_prism_bwd(Cj, jt, gp, sign);
        }
      }
    }
    /* centre of mass: kinematics_backward_helper.cuh:186-291 */
    if (compute_com && grad_com && batch_com) {
      const float total_mass = batch_com[(size_t)n * 4 + 3];
      const float *gc = grad_com + (size_t)n * 4;
      if (total_mass > 0.0f && !(gc[0] == 0 && gc[1] == 0 && gc[2] == 0)) {
        for (int l = 0; l < nlinks; l++) {
          const float *mc = link_masses_com + l * 4;
          const float mass = mc[3];
          if (mass <= 0.0f) continue;
          const float g[3] = {gc[0] * mass / total_mass, gc[1] * mass / total_mass,
                              gc[2] * mass / total_mass};
          float cw[4];
          orc_transform_sphere(cumul + l * 12, mc, cw);
          for (int ci = link_chain_offsets[l + 1] - 1; ci >= link_chain_offsets[l]; ci--) {
            const int j = link_chain_data[ci];
            const int jt = joint_map_type[j];
            const float sign = joint_offset[j * 2];
            if (jt >= J_X_ROT && jt <= J_Z_ROT)
              psum[joint_map[j]] += orc_rot_bwd_translation(cumul + j * 12, jt - J_X_ROT, cw, g, sign);
            else if (jt >= J_X_PRISM && jt <= J_Z_PRISM)
              psum[joint_map[j]] += orc_prism_bwd(cumul + j * 12, jt, g, sign);
          }
        }
      }
    }
    for (int j = 0; j < njoints; j++) grad_q[(size_t)n * njoints + j] = psum[j];
  }
}

/* ------------------------------------------------------------------------------------------
 * A3. self collision (self_collision_kernel.cuh:19-297, self_collision_helper.cuh:61-349)
 *     Result of the single-block and of the map-reduce variant is the same function of the
 *     inputs, so one restatement covers both (num_blocks_per_batch only changes the launch).
 *     out_gradient / sparse_index keep the reference's stateful contract: rows flagged by the
 *     previous call are zeroed first, then the arg-max pair's two rows are written.
 *     out_pair_idx[N,2] (oracle-only extra): the arg-max pair, (-1,-1) when no collision.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_self_collision(float *out_distance, float *out_gradient, float *pair_distance,
                                uint8_t *sparse_index, int16_t *out_pair_idx,
                                const float *robot_spheres, const float *offsets,
                                const float *weight, const int16_t *pair_locations, int n_points,
                                int nspheres, int npairs, int store_pair_distance, int write_grad) {
  const float w = weight[0];
#pragma omp parallel
  {
    float *sph = (float *)malloc((size_t)nspheres * 4 * sizeof(float));
#pragma omp for schedule(static)
    for (int n = 0; n < n_points; n++) {
      /* load_spheres_and_zero_gradients: self_collision_helper.cuh:151-192 */
      for (int s = 0; s < nspheres; s++) {
        const float *src = robot_spheres + ((size_t)n * nspheres + s) * 4;
        sph[s * 4 + 0] = src[0]; sph[s * 4 + 1] = src[1]; sph[s * 4 + 2] = src[2];
        sph[s * 4 + 3] = src[3] + offsets[s];
        if (sparse_index && sparse_index[(size_t)n * nspheres + s]) {
          float *g = out_gradient + ((size_t)n * nspheres + s) * 4;
          g[0] = g[1] = g[2] = g[3] = 0.0f;
          sparse_index[(size_t)n * nspheres + s] = 0;
        }
      }
      /* compute_max_collision_distance: :226-275 with the canonical lowest-index tie rule */
      float best = 0.0f;
      int bi = 0, bj = 0, found = 0;
      for (int p = 0; p < npairs; p++) {
        const int i = pair_locations[2 * p], j = pair_locations[2 * p + 1];
        const float *a = sph + i * 4, *b = sph + j * 4;
        const float valid = (a[3] >= 0.0f && b[3] >= 0.0f) ? 1.0f : 0.0f;
        /* sphere_squared_distance_fused: :61-71 */
        const float r = a[3] + b[3];
        const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        const float f = ((r * r) - d2) * valid;
        if (store_pair_distance && pair_distance) pair_distance[(size_t)n * npairs + p] = f;
        if (f > best) { best = f; bi = i; bj = j; found = 1; }
      }
      /* finalize_collision_results: :277-349 */
      if (out_pair_idx) { out_pair_idx[2 * n] = -1; out_pair_idx[2 * n + 1] = -1; }
      if (!found || best <= 0.0f) { out_distance[n] = 0.0f; continue; }
      out_distance[n] = 0.5f * w * best;
      if (out_pair_idx) { out_pair_idx[2 * n] = (int16_t)bi; out_pair_idx[2 * n + 1] = (int16_t)bj; }
      if (write_grad) {
        const float *a = sph + bi * 4, *b = sph + bj * 4;
        float *gi = out_gradient + ((size_t)n * nspheres + bi) * 4;
        float *gj = out_gradient + ((size_t)n * nspheres + bj) * 4;
        const float vx = w * (b[0] - a[0]), vy = w * (b[1] - a[1]), vz = w * (b[2] - a[2]);
        gi[0] = vx; gi[1] = vy; gi[2] = vz; gi[3] = w * -1.0f;
        gj[0] = -1.0f * vx; gj[1] = -1.0f * vy; gj[2] = -1.0f * vz; gj[3] = w * -1.0f;
        sparse_index[(size_t)n * nspheres + bi] = 1;
        sparse_index[(size_t)n * nspheres + bj] = 1;
      }
    }
    free(sph);
  }
}

/* ------------------------------------------------------------------------------------------
 * A4. sphere--scene collision (Warp in the reference)
 *   geom/collision/wp_collision_kernel.py:70-166, wp_sweep_collision_kernel.py:83-260,
 *   wp_collision_common.py:11-38, geom/data/data_cuboid.py:547-628, data_voxel.py:709-1215,
 *   geom/data/helper_pose.py:13-90.  Warp intrinsics (transform_point, quat_rotate,
 *   transform_inverse) are restated from their definitions (warp-lang is not in the tree,
 *   pyproject.toml:38): quat_rotate(q,v) = v(2w^2-1) + 2w(q x v) + 2q(q.v).
 *   Pinned by the reference's own kernel sources run on the CPU through the Warp stand-in of
 *   tests/golden/warp_emulator (tests/golden/scene_warp_golden.npz: bit-equal distances).
 *   Obstacle sums are accumulated in obstacle-index order (the reference uses float atomics
 *   whose order is undefined), cuboids first then voxel grids.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float p[3]; float q[4]; /* xyzw */ float m[3][4]; /* mode 1 only: rows of [R | p] */ } orc_tf;

/* Arithmetic of the world -> obstacle-frame transform of the sphere centres.
 *   0 (default): the reference's -- Warp's quat_rotate + translation, as restated above.
 *   1: the rotation-matrix form of the HIP path (curobo_amd/csrc/scene_device.hpp::load_rec_global + to_local: nine
 *      entries from explicit fused multiply-adds, then a nest of three per coordinate).  Both round the same exact
 *      transform; they differ in the last bit of a local coordinate.  The ONLY place of the algorithm where that bit is
 *      a decision is the sweep's `jump >= half_dist` at jump = 0 (wp_sweep_collision_kernel.py:186-203): a sphere whose
 *      world positions at h and h +- 1 differ by an ulp coincides with its neighbour in one arithmetic's obstacle frame
 *      and not in the other's, and then gets one more copy of its centre sample.  Mode 1 lets a test feed the oracle
 *      the device's own spheres and compare EVERY trajectory tightly; mode 0 against mode 1 on the same spheres isolates
 *      exactly those resting spheres (tests/test_oracle_collision.py).  Gradients go back to the world frame with the
 *      reference's quaternion form in both modes. */
static int orc_frame_arithmetic = 0;
ORC_API void orc_set_frame_arithmetic(int mode) { orc_frame_arithmetic = mode; }
ORC_API int orc_get_frame_arithmetic(void) { return orc_frame_arithmetic; }

static void orc_quat_rotate(const float *q, const float *v, float *o) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float cx = y * v[2] - z * v[1], cy = z * v[0] - x * v[2], cz = x * v[1] - y * v[0];
  const float d = x * v[0] + y * v[1] + z * v[2];
  const float k = 2.0f * w * w - 1.0f;
  o[0] = v[0] * k + cx * w * 2.0f + x * d * 2.0f;
  o[1] = v[1] * k + cy * w * 2.0f + y * d * 2.0f;
  o[2] = v[2] * k + cz * w * 2.0f + z * d * 2.0f;
}

static void orc_load_inv_tf(const float *inv_pose8, orc_tf *t) {
  /* helper_pose.py:28-90: [x y z qw qx qy qz pad] -> wp.transform(pos, quat(x,y,z,w)) */
  t->p[0] = inv_pose8[0]; t->p[1] = inv_pose8[1]; t->p[2] = inv_pose8[2];
  t->q[0] = inv_pose8[4]; t->q[1] = inv_pose8[5]; t->q[2] = inv_pose8[6]; t->q[3] = inv_pose8[3];
  if (orc_frame_arithmetic == 1) {
    const float x = t->q[0], y = t->q[1], z = t->q[2], w = t->q[3];
    const float x2 = 2.0f * x, y2 = 2.0f * y, w2 = 2.0f * w;
    const float k = __builtin_fmaf(w2, w, -1.0f);
    const float wz = w2 * z, wy = w2 * y, wx = w2 * x;
    t->m[0][0] = __builtin_fmaf(x2, x, k);   t->m[0][1] = __builtin_fmaf(x2, y, -wz); t->m[0][2] = __builtin_fmaf(x2, z, wy);
    t->m[1][0] = __builtin_fmaf(x2, y, wz);  t->m[1][1] = __builtin_fmaf(y2, y, k);   t->m[1][2] = __builtin_fmaf(y2, z, -wx);
    t->m[2][0] = __builtin_fmaf(x2, z, -wy); t->m[2][1] = __builtin_fmaf(y2, z, wx);  t->m[2][2] = __builtin_fmaf(2.0f * z, z, k);
    t->m[0][3] = t->p[0]; t->m[1][3] = t->p[1]; t->m[2][3] = t->p[2];
  }
}

static void orc_tf_point(const orc_tf *t, const float *v, float *o) {
  if (orc_frame_arithmetic == 1) {
    for (int r = 0; r < 3; r++)
      o[r] = __builtin_fmaf(t->m[r][0], v[0], __builtin_fmaf(t->m[r][1], v[1], __builtin_fmaf(t->m[r][2], v[2], t->m[r][3])));
    return;
  }
  orc_quat_rotate(t->q, v, o);
  o[0] += t->p[0]; o[1] += t->p[1]; o[2] += t->p[2];
}

/* rotate a local vector back to the world frame: transform_vector(transform_inverse(t), v) */
static void orc_tf_inv_vector(const orc_tf *t, const float *v, float *o) {
  const float qi[4] = {-t->q[0], -t->q[1], -t->q[2], t->q[3]};
  orc_quat_rotate(qi, v, o);
}

/* wp_collision_common.py:11-38 */
static void orc_activation(float dist, float eta, float *cost, float *gscale) {
  if (dist <= 0.0f) { *cost = 0.0f; *gscale = 0.0f; return; }
  if (dist > eta) { *cost = dist - 0.5f * eta; *gscale = 1.0f; }
  else { *cost = 0.5f * dist * dist / eta; *gscale = dist / eta; }
}

/* Analytic primitives of the cuboid store (an extension over the reference, which meshes them: geom/types.py
 * :290-450, :1104-1124): dims[3] = 1 sphere (dims[0] = radius), 2 capsule (radius, half segment length on local z),
 * 3 cylinder (radius, half height on local z).  Closed forms; g = minus the SDF gradient. */
static float orc_primitive_sdf(const float *dims, const float *lp, float *g) {
  const int tag = (int)dims[3];
  const float r = dims[0], hl = tag == 1 ? 0.0f : dims[1];
  g[0] = g[1] = g[2] = 0.0f;
  if (tag == 3) {
    const float rho = sqrtf(lp[0] * lp[0] + lp[1] * lp[1]);
    const float dr = rho - r, dz = fabsf(lp[2]) - hl;
    const float cr = fmaxf(dr, 0.0f), cz = fmaxf(dz, 0.0f);
    const float od = sqrtf(cr * cr + cz * cz);
    const float sdf = od + fminf(fmaxf(dr, dz), 0.0f);
    const float ex = rho > 1e-6f ? lp[0] / rho : 1.0f, ey = rho > 1e-6f ? lp[1] / rho : 0.0f;
    const float sz = lp[2] < 0.0f ? -1.0f : 1.0f;
    if (od > 1e-6f) { g[0] = -ex * cr / od; g[1] = -ey * cr / od; g[2] = -sz * cz / od; }
    else if (dr > dz) { g[0] = -ex; g[1] = -ey; }
    else g[2] = -sz;
    return sdf;
  }
  const float t = fminf(fmaxf(lp[2], -hl), hl);
  const float vx = lp[0], vy = lp[1], vz = lp[2] - t;
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  if (d > 1e-6f) { g[0] = -vx / d; g[1] = -vy / d; g[2] = -vz / d; }
  else g[2] = -1.0f;
  return d - r;
}

/* data_cuboid.py:547-628; returns sdf, g = minus the SDF gradient (cost gradient direction) */
static float orc_cuboid_sdf(const float *dims, const float *lp, float *g) {
  if (dims[3] != 0.0f) return orc_primitive_sdf(dims, lp, g);
  const float hx = dims[0] * 0.5f, hy = dims[1] * 0.5f, hz = dims[2] * 0.5f;
  const float qx = fabsf(lp[0]) - hx, qy = fabsf(lp[1]) - hy, qz = fabsf(lp[2]) - hz;
  const float cx = fmaxf(qx, 0.0f), cy = fmaxf(qy, 0.0f), cz = fmaxf(qz, 0.0f);
  const float od = sqrtf(cx * cx + cy * cy + cz * cz);
  const float sdf = od + fminf(fmaxf(qx, fmaxf(qy, qz)), 0.0f);
  g[0] = g[1] = g[2] = 0.0f;
  if (od > 1e-6f) {
    const float inv = -1.0f / od;
    g[0] = cx * inv; g[1] = cy * inv; g[2] = cz * inv;
    if (lp[0] < 0.0f) g[0] = -g[0];
    if (lp[1] < 0.0f) g[1] = -g[1];
    if (lp[2] < 0.0f) g[2] = -g[2];
  } else {
    const float mq = fmaxf(qx, fmaxf(qy, qz));
    if (fabsf(qx - mq) < 1e-6f) g[0] = (lp[0] < 0.0f) ? 1.0f : -1.0f;
    else if (fabsf(qy - mq) < 1e-6f) g[1] = (lp[1] < 0.0f) ? 1.0f : -1.0f;
    else g[2] = (lp[2] < 0.0f) ? 1.0f : -1.0f;
  }
  return sdf;
}

/* IEEE binary16 -> binary32 */
static float orc_half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      exp = 127 - 15 + 1;
      while ((man & 0x400u) == 0) { man <<= 1; exp--; }
      man &= 0x3ffu;
      bits = sign | (exp << 23) | (man << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

typedef struct {
  /* cuboids: data_cuboid.py:67-108 */
  const float *cub_dims;      /* [num_envs, max_cub, 4] */
  const float *cub_inv_pose;  /* [num_envs, max_cub, 8] */
  const uint8_t *cub_enable;  /* [num_envs, max_cub] */
  const int32_t *cub_count;   /* [num_envs] */
  int max_cub;
  /* voxel grids: data_voxel.py:42-95 */
  const float *vox_params;    /* [num_envs, max_vox, 4] = voxel counts nx ny nz, voxel_size */
  const float *vox_inv_pose;  /* [num_envs, max_vox, 8] */
  const uint8_t *vox_enable;  /* [num_envs, max_vox] */
  const int32_t *vox_count;   /* [num_envs] */
  const uint16_t *vox_features; /* fp16 [num_envs, max_vox, n_voxels] */
  int max_vox;
  int vox_n_voxels;           /* per-grid feature stride */
  float vox_max_distance;
  /* meshes: data_mesh.py:60-120 (MeshData); the triangles of every loaded mesh, concatenated.  No BVH here: the oracle
   * walks all triangles of a mesh for every query (exact by construction). */
  const float *mesh_dims;       /* [num_envs, max_mesh, 4] bounding-box extents */
  const float *mesh_inv_pose;   /* [num_envs, max_mesh, 8] */
  const uint8_t *mesh_enable;   /* [num_envs, max_mesh] */
  const int32_t *mesh_count;    /* [num_envs] */
  const int32_t *mesh_id;       /* [num_envs, max_mesh] index of the loaded mesh */
  const float *mesh_vertices;   /* [sum V, 3] */
  const int32_t *mesh_faces;    /* [sum F, 3], indices local to the mesh */
  const int32_t *mesh_vert_offset, *mesh_face_offset; /* [n_meshes + 1] */
  int max_mesh;
} orc_scene;

/* ---- triangle meshes: data_mesh.py:630-700 (compute_local_sdf_with_grad) with the closest point found by brute force
 * (Ericson, Real-Time Collision Detection 5.1.5).  The SIGN of the query is Warp's (wp.mesh_query_point, data_mesh.py:632,
 * 682; warp-lang >= 0.10.0 is the reference's pin, pyproject.toml:37; the package is not in /root/reference and has no
 * ROCm build, so its PUBLISHED algorithm is restated, warp/native/mesh.h):
 *   rule 1, "rays" = mesh_query_point -> mesh_query_inside: three rays from the query point along +x, +y and +z; each ray's
 *     NEAREST hit (mesh_query_ray, t > 0) reports on which side of the hit face the ray origin lies ("sign > 0 if the ray
 *     hit in front of the face", i.e. dot(direction, ab x ac) < 0); the point is INSIDE iff all three rays hit and all three
 *     nearest hits are back faces, else outside.  Restated in double precision over every triangle.
 *   rule 0, "winding" (the default the oracle has had since round 3) = the generalised winding number (van Oosterom-
 *     Strackee solid angles summed in double precision), |w| > 0.5 = inside.
 * On a closed, consistently oriented surface the two rules and the closest-feature pseudonormal rule of the HIP path
 * (Baerentzen & Aanaes) are the same function away from the surface; on open or inconsistently oriented meshes they are three
 * different functions and only rule 1 is the reference's (tests/test_oracle_mesh_sign.py tabulates where they part). */
static int orc_mesh_sign_rule = 0;
ORC_API void orc_set_mesh_sign_rule(int rule) { orc_mesh_sign_rule = rule; }
ORC_API int orc_get_mesh_sign_rule(void) { return orc_mesh_sign_rule; }

/* mesh_query_inside (warp/native/mesh.h): 1 = inside.  Moeller-Trumbore in double precision; a ray that passes exactly
 * through an edge or a vertex is a degenerate case of Warp's fp32 watertight test too -- the fixtures avoid them. */
static int orc_mesh_inside_rays(const float *verts, const int32_t *faces, int n_faces, const float *lp) {
  int votes = 0;
  for (int axis = 0; axis < 3; axis++) {
    double best_t = 1e300;
    int best_back = 0, hit = 0;
    for (int f = 0; f < n_faces; f++) {
      const float *a = verts + (size_t)faces[f * 3] * 3, *b = verts + (size_t)faces[f * 3 + 1] * 3, *c = verts + (size_t)faces[f * 3 + 2] * 3;
      double ab[3], ac[3], tv[3], d[3] = {0.0, 0.0, 0.0};
      d[axis] = 1.0;
      for (int i = 0; i < 3; i++) { ab[i] = (double)b[i] - a[i]; ac[i] = (double)c[i] - a[i]; tv[i] = (double)lp[i] - a[i]; }
      const double pv[3] = {d[1] * ac[2] - d[2] * ac[1], d[2] * ac[0] - d[0] * ac[2], d[0] * ac[1] - d[1] * ac[0]};
      const double det = ab[0] * pv[0] + ab[1] * pv[1] + ab[2] * pv[2];
      if (det == 0.0) continue;
      const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
      if (u < 0.0 || u > 1.0) continue;
      const double qv[3] = {tv[1] * ab[2] - tv[2] * ab[1], tv[2] * ab[0] - tv[0] * ab[2], tv[0] * ab[1] - tv[1] * ab[0]};
      const double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det;
      if (v < 0.0 || u + v > 1.0) continue;
      const double t = (ac[0] * qv[0] + ac[1] * qv[1] + ac[2] * qv[2]) / det;
      if (t <= 0.0 || t >= best_t) continue;
      /* the face normal ab x ac against the ray: d . n = det' -- with pv = d x ac, det = ab . (d x ac) = -d . (ab x ac) */
      best_t = t; hit = 1; best_back = det < 0.0;  /* d . n > 0: the ray leaves through the back of the face */
    }
    if (hit && best_back) votes++;
  }
  return votes == 3;
}

static void orc_closest_on_triangle(const float *p, const float *a, const float *b, const float *c, float *out) {
  float ab[3], ac[3], ap[3];
  for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
#define ORC_DOT(x, y) ((x)[0] * (y)[0] + (x)[1] * (y)[1] + (x)[2] * (y)[2])
  const float d1 = ORC_DOT(ab, ap), d2 = ORC_DOT(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) { for (int i = 0; i < 3; i++) out[i] = a[i]; return; }
  float bp[3];
  for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
  const float d3 = ORC_DOT(ab, bp), d4 = ORC_DOT(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) { for (int i = 0; i < 3; i++) out[i] = b[i]; return; }
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    const float v = d1 / (d1 - d3);
    for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i];
    return;
  }
  float cp[3];
  for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
  const float d5 = ORC_DOT(ab, cp), d6 = ORC_DOT(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) { for (int i = 0; i < 3; i++) out[i] = c[i]; return; }
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    const float w = d2 / (d2 - d6);
    for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i];
    return;
  }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
    return;
  }
  const float den = 1.0f / (va + vb + vc);
  for (int i = 0; i < 3; i++) out[i] = a[i] + (vb * den) * ab[i] + (vc * den) * ac[i];
}

/* signed distance (negative inside) and gradient (p - closest) / |p - closest| of one mesh; max_distance when no surface
 * lies within max_distance (data_mesh.py:670-672) */
static float orc_mesh_sdf_raw(const float *verts, const int32_t *faces, int n_faces, const float *lp, float max_distance, float *g) {
  g[0] = g[1] = g[2] = 0.0f;
  float best2 = max_distance * max_distance, cl[3] = {0, 0, 0};
  int found = 0;
  double solid = 0.0;
  for (int f = 0; f < n_faces; f++) {
    const float *a = verts + (size_t)faces[f * 3] * 3, *b = verts + (size_t)faces[f * 3 + 1] * 3, *c = verts + (size_t)faces[f * 3 + 2] * 3;
    float q[3];
    orc_closest_on_triangle(lp, a, b, c, q);
    const float dx = lp[0] - q[0], dy = lp[1] - q[1], dz = lp[2] - q[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    if (d2 <= best2) { best2 = d2; cl[0] = q[0]; cl[1] = q[1]; cl[2] = q[2]; found = 1; }
    double ra[3], rb[3], rc[3];
    for (int i = 0; i < 3; i++) { ra[i] = (double)a[i] - lp[i]; rb[i] = (double)b[i] - lp[i]; rc[i] = (double)c[i] - lp[i]; }
    const double la = sqrt(ORC_DOT(ra, ra)), lb = sqrt(ORC_DOT(rb, rb)), lc = sqrt(ORC_DOT(rc, rc));
    const double cr[3] = {rb[1] * rc[2] - rb[2] * rc[1], rb[2] * rc[0] - rb[0] * rc[2], rb[0] * rc[1] - rb[1] * rc[0]};
    const double num = ORC_DOT(ra, cr);
    const double den = la * lb * lc + ORC_DOT(ra, rb) * lc + ORC_DOT(rb, rc) * la + ORC_DOT(rc, ra) * lb;
    solid += 2.0 * atan2(num, den);
  }
#undef ORC_DOT
  if (!found) return max_distance;
  const float d = sqrtf(best2);
  if (d > 1e-6f) { g[0] = (lp[0] - cl[0]) / d; g[1] = (lp[1] - cl[1]) / d; g[2] = (lp[2] - cl[2]) / d; }
  const int inside = orc_mesh_sign_rule == 1 ? orc_mesh_inside_rays(verts, faces, n_faces, lp) : fabs(solid) > 6.283185307179586;
  return inside ? -d : d;
}

ORC_API void orc_mesh_query(float *out_sdf, float *out_grad, const float *points, const float *vertices, const int32_t *faces,
                            int n_faces, float max_distance, int n_points) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_points; i++) out_sdf[i] = orc_mesh_sdf_raw(vertices, faces, n_faces, points + (size_t)i * 3, max_distance, out_grad + (size_t)i * 3);
}

/* Voxel ESDF lookup, restating data_voxel.py (trilinear, align-corners, analytic gradient).
 * See orc_voxel_sdf() below -- defined after the helper for clarity. */
static float orc_voxel_sdf(const orc_scene *sc, int flat_idx, const float *lp, float *g,
                           int *valid);

static void orc_scene_eval_point(const orc_scene *sc, int env, int is_voxel, int o,
                                 const float *lp, float r_adj, float eta, float *cost_sum,
                                 float *grad_sum, float *pen_out) {
  float g[3];
  float sdf;
  if (is_voxel == 0) {
    sdf = orc_cuboid_sdf(sc->cub_dims + ((size_t)env * sc->max_cub + o) * 4, lp, g);
  } else if (is_voxel == 1) {
    int valid = 1;
    sdf = orc_voxel_sdf(sc, env * sc->max_vox + o, lp, g, &valid);
  } else {  /* mesh: data_mesh.py:630-700; max_distance = max(half the bounding-box diagonal, the query distance) */
    const size_t flat = (size_t)env * sc->max_mesh + o;
    const float *dm = sc->mesh_dims + flat * 4;
    float max_distance = 0.5f * sqrtf(dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2]);
    if (r_adj > max_distance) max_distance = r_adj;
    const int m = sc->mesh_id[flat];
    sdf = orc_mesh_sdf_raw(sc->mesh_vertices + (size_t)sc->mesh_vert_offset[m] * 3, sc->mesh_faces + (size_t)sc->mesh_face_offset[m] * 3,
                           sc->mesh_face_offset[m + 1] - sc->mesh_face_offset[m], lp, max_distance, g);
  }
  const float pen = -sdf + r_adj;
  *pen_out = pen;
  if (pen > 0.0f) {
    float c, gs;
    orc_activation(pen, eta, &c, &gs);
    *cost_sum += c;
    grad_sum[0] += gs * g[0]; grad_sum[1] += gs * g[1]; grad_sum[2] += gs * g[2];
  }
}

/*
 * orc_scene_collision: SphereObstacleCollision / SweptSphereObstacleCollision forward
 * (geom/collision/wp_autograd.py:37-249).  distance[N*S], gradient[N*S*4] are fully
 * rewritten (the reference zero_()s them first, :76,:176).
 *   sweep_steps = 0 -> discrete kernel; 3 -> swept kernel (SWEEP_STEPS, :66).
 *   enable_speed_metric applies wp_speed_metric.py:10-93 afterwards; speed_dt is the
 *   reference's single-element tensor.
 */
ORC_API void orc_scene_collision(float *distance, float *gradient, const float *spheres,
                                 const orc_scene *sc, const float *weight,
                                 const float *activation_distance, const int32_t *env_query_idx,
                                 int batch, int horizon, int nspheres, int use_multi_env,
                                 int sweep_steps, int enable_speed_metric, const float *speed_dt) {
  const float w = weight[0], eta = activation_distance[0];
  const long total = (long)batch * horizon * nspheres;
#pragma omp parallel for schedule(static)
  for (long sidx = 0; sidx < total; sidx++) {
    const int b = (int)(sidx / ((long)horizon * nspheres));
    const int h = (int)((sidx - (long)b * horizon * nspheres) / nspheres);
    const int env = use_multi_env ? env_query_idx[b] : 0;
    const float *s = spheres + sidx * 4;
    float dsum = 0.0f, gsum[3] = {0, 0, 0};
    if (s[3] >= 0.0f) {
      const float r_adj = s[3] + eta;
      /* the reference launches its kernel once per obstacle kind: cuboids, meshes, voxels (data_scene.py:72-81); sums of
       * floats: the kinds are visited cuboids, voxels, meshes here and in the HIP path alike */
      for (int kind = 0; kind < 3; kind++) {
        const int max_n = kind == 0 ? sc->max_cub : (kind == 1 ? sc->max_vox : sc->max_mesh);
        if (max_n <= 0) continue;
        const int32_t *count = kind == 0 ? sc->cub_count : (kind == 1 ? sc->vox_count : sc->mesh_count);
        const uint8_t *enable = kind == 0 ? sc->cub_enable : (kind == 1 ? sc->vox_enable : sc->mesh_enable);
        const float *inv_pose = kind == 0 ? sc->cub_inv_pose : (kind == 1 ? sc->vox_inv_pose : sc->mesh_inv_pose);
        for (int o = 0; o < max_n; o++) {
          /* is_obs_enabled: data_cuboid.py:467-485 */
          if (o >= count[env]) continue;
          if (enable[(size_t)env * max_n + o] != 1) continue;
          orc_tf t;
          orc_load_inv_tf(inv_pose + ((size_t)env * max_n + o) * 8, &t);
          float lc[3];
          orc_tf_point(&t, s, lc);
          float cost_sum = 0.0f, grad_local[3] = {0, 0, 0}, pen;
          orc_scene_eval_point(sc, env, kind, o, lc, r_adj, eta, &cost_sum, grad_local, &pen);
          /* sweeps: wp_sweep_collision_kernel.py:176-254 */
          for (int dir = 0; dir < 2 && sweep_steps > 0; dir++) {
            if (dir == 0 && !(h > 0)) continue;
            if (dir == 1 && !(h < horizon - 1)) continue;
            const float *ns = (dir == 0) ? s - (long)nspheres * 4 : s + (long)nspheres * 4;
            float ln[3];
            orc_tf_point(&t, ns, ln);
            const float ddx = ln[0] - lc[0], ddy = ln[1] - lc[1], ddz = ln[2] - lc[2];
            const float half_dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * 0.5f;
            const float inv_half = 1.0f / fmaxf(half_dist, 0.001f);
            float jump = 0.0f;
            for (int k = 0; k < sweep_steps; k++) {
              if (jump >= half_dist) break;
              const float tt = 1.0f - 0.5f * jump * inv_half;
              const float lp[3] = {tt * lc[0] + (1.0f - tt) * ln[0],
                                   tt * lc[1] + (1.0f - tt) * ln[1],
                                   tt * lc[2] + (1.0f - tt) * ln[2]};
              float p2;
              orc_scene_eval_point(sc, env, kind, o, lp, r_adj, eta, &cost_sum, grad_local, &p2);
              if (p2 > 0.0f) jump += p2;
              else if (-p2 >= 1000.0f) jump += r_adj;
              else jump += fmaxf(-p2, r_adj);
            }
          }
          if (cost_sum > 0.0f) {
            float gw[3];
            orc_tf_inv_vector(&t, grad_local, gw);
            dsum += w * cost_sum;
            gsum[0] += w * gw[0]; gsum[1] += w * gw[1]; gsum[2] += w * gw[2];
          }
        }
      }
    }
    distance[sidx] = dsum;
    gradient[sidx * 4 + 0] = gsum[0]; gradient[sidx * 4 + 1] = gsum[1];
    gradient[sidx * 4 + 2] = gsum[2]; gradient[sidx * 4 + 3] = 0.0f;
  }
  if (!enable_speed_metric) return;
  /* wp_speed_metric.py:38-93 (reads only spheres + this sphere's own accumulators) */
  float dt = speed_dt[0];
  if (dt < 1e-6f) dt = 1e-6f;
#pragma omp parallel for schedule(static)
  for (long sidx = 0; sidx < total; sidx++) {
    const int b = (int)(sidx / ((long)horizon * nspheres));
    const int h = (int)((sidx - (long)b * horizon * nspheres) / nspheres);
    if (h == 0 || h >= horizon - 1) continue;
    const float *pp = spheres + (sidx - nspheres) * 4;
    const float *cp = spheres + sidx * 4;
    const float *np = spheres + (sidx + nspheres) * 4;
    const float k = 0.5f / dt;
    const float vel[3] = {k * (np[0] - pp[0]), k * (np[1] - pp[1]), k * (np[2] - pp[2])};
    const float sv = sqrtf(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
    if (sv < 1e-3f) continue;
    const float d = distance[sidx];
    if (d <= 0.0f) continue;
    const float g[3] = {gradient[sidx * 4], gradient[sidx * 4 + 1], gradient[sidx * 4 + 2]};
    const float ka = 1.0f / (dt * dt);
    const float acc[3] = {ka * (pp[0] + np[0] - 2.0f * cp[0]), ka * (pp[1] + np[1] - 2.0f * cp[1]),
                          ka * (pp[2] + np[2] - 2.0f * cp[2])};
    const float nv[3] = {vel[0] / sv, vel[1] / sv, vel[2] / sv};
    const float sv2 = sv * sv;
    const float curv[3] = {acc[0] / sv2, acc[1] / sv2, acc[2] / sv2};
    const float dg = nv[0] * g[0] + nv[1] * g[1] + nv[2] * g[2];
    const float dc = nv[0] * curv[0] + nv[1] * curv[1] + nv[2] * curv[2];
    for (int a = 0; a < 3; a++) {
      const float og = g[a] - dg * nv[a];
      const float oc = curv[a] - dc * nv[a];
      gradient[sidx * 4 + a] = sv * (og - d * oc);
    }
    distance[sidx] = sv * d;
  }
}

/* data_voxel.py:781-1056 (sample_voxel_sdf_with_grad) + :1164-1215 (compute_local_sdf_with_grad).
 * g returns the normalised NEGATIVE SDF gradient (cost direction), zero when sdf >= max_dist. */
static float orc_voxel_sdf(const orc_scene *sc, int flat_idx, const float *lp, float *g,
                           int *valid) {
  const float *prm = sc->vox_params + (size_t)flat_idx * 4;
  const int nx = (int)prm[0], ny = (int)prm[1], nz = (int)prm[2];
  const float vs = prm[3];
  const float max_dist = sc->vox_max_distance;
  const uint16_t *feat = sc->vox_features + (size_t)flat_idx * sc->vox_n_voxels;
  float sdf, gx = 0.0f, gy = 0.0f, gz = 0.0f;
  *valid = 1;
  if (nx < 2 || ny < 2 || nz < 2) {
    /* world_to_voxel_idx: data_voxel.py:709-724 */
    const int ix = (int)((lp[0] + (float)nx * vs * 0.5f) / vs);
    const int iy = (int)((lp[1] + (float)ny * vs * 0.5f) / vs);
    const int iz = (int)((lp[2] + (float)nz * vs * 0.5f) / vs);
    if (ix < 0 || ix >= nx || iy < 0 || iy >= ny || iz < 0 || iz >= nz) sdf = max_dist;
    else sdf = orc_half_to_float(feat[ix * ny * nz + iy * nz + iz]);
  } else {
    const float inv_voxel = 1.0f / vs;
    const float vx = lp[0] * inv_voxel + (float)nx * 0.5f - 0.5f;
    const float vy = lp[1] * inv_voxel + (float)ny * 0.5f - 0.5f;
    const float vz = lp[2] * inv_voxel + (float)nz * 0.5f - 0.5f;
    const int x0 = (int)floorf(vx), y0 = (int)floorf(vy), z0 = (int)floorf(vz);
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float fx = vx - (float)x0, fy = vy - (float)y0, fz = vz - (float)z0;
    const float fx1 = 1.0f - fx, fy1 = 1.0f - fy, fz1 = 1.0f - fz;
    const int sx = ny * nz, sy = nz;
    const int x0ok = x0 >= 0 && x0 < nx, x1ok = x1 >= 0 && x1 < nx;
    const int y0ok = y0 >= 0 && y0 < ny, y1ok = y1 >= 0 && y1 < ny;
    const int z0ok = z0 >= 0 && z0 < nz, z1ok = z1 >= 0 && z1 < nz;
    const long base = (long)x0 * sx + (long)y0 * sy + z0;
    float s[8], v[8];
    const int ok[8] = {x0ok && y0ok && z0ok, x0ok && y0ok && z1ok, x0ok && y1ok && z0ok,
                       x0ok && y1ok && z1ok, x1ok && y0ok && z0ok, x1ok && y0ok && z1ok,
                       x1ok && y1ok && z0ok, x1ok && y1ok && z1ok};
    const long off[8] = {0, 1, sy, sy + 1, sx, sx + 1, sx + sy, sx + sy + 1};
    int all_valid = 1;
    for (int k = 0; k < 8; k++) {
      if (ok[k]) { s[k] = orc_half_to_float(feat[base + off[k]]); v[k] = 1.0f; }
      else { s[k] = max_dist; v[k] = 0.0f; all_valid = 0; }
    }
    /* index k = (x<<2)|(y<<1)|z : s000 s001 s010 s011 s100 s101 s110 s111 */
    if (all_valid) {
      sdf = s[0] * fx1 * fy1 * fz1 + s[1] * fx1 * fy1 * fz + s[2] * fx1 * fy * fz1 +
            s[3] * fx1 * fy * fz + s[4] * fx * fy1 * fz1 + s[5] * fx * fy1 * fz +
            s[6] * fx * fy * fz1 + s[7] * fx * fy * fz;
      gx = ((s[4] - s[0]) * fy1 * fz1 + (s[5] - s[1]) * fy1 * fz + (s[6] - s[2]) * fy * fz1 +
            (s[7] - s[3]) * fy * fz) * inv_voxel;
      gy = ((s[2] - s[0]) * fx1 * fz1 + (s[3] - s[1]) * fx1 * fz + (s[6] - s[4]) * fx * fz1 +
            (s[7] - s[5]) * fx * fz) * inv_voxel;
      gz = ((s[1] - s[0]) * fx1 * fy1 + (s[3] - s[2]) * fx1 * fy + (s[5] - s[4]) * fx * fy1 +
            (s[7] - s[6]) * fx * fy) * inv_voxel;
    } else {
      const float wts[8] = {fx1 * fy1 * fz1, fx1 * fy1 * fz, fx1 * fy * fz1, fx1 * fy * fz,
                            fx * fy1 * fz1,  fx * fy1 * fz,  fx * fy * fz1,  fx * fy * fz};
      float wsum = 0.0f, vsum = 0.0f;
      for (int k = 0; k < 8; k++) { vsum += s[k] * wts[k] * v[k]; wsum += wts[k] * v[k]; }
      if (wsum <= 0.0f) {
        sdf = max_dist;
      } else {
        sdf = vsum / wsum;
        /* pairs (lo,hi) and their bilinear weights, data_voxel.py:982-1049 */
        const int px[4][2] = {{0, 4}, {1, 5}, {2, 6}, {3, 7}};
        const float wx[4] = {fy1 * fz1, fy1 * fz, fy * fz1, fy * fz};
        const int py[4][2] = {{0, 2}, {1, 3}, {4, 6}, {5, 7}};
        const float wy[4] = {fx1 * fz1, fx1 * fz, fx * fz1, fx * fz};
        const int pz[4][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}};
        const float wz[4] = {fx1 * fy1, fx1 * fy, fx * fy1, fx * fy};
        float gs, gw;
        gs = gw = 0.0f;
        for (int k = 0; k < 4; k++)
          if (v[px[k][0]] > 0.0f && v[px[k][1]] > 0.0f) { gs += (s[px[k][1]] - s[px[k][0]]) * wx[k]; gw += wx[k]; }
        gx = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
        gs = gw = 0.0f;
        for (int k = 0; k < 4; k++)
          if (v[py[k][0]] > 0.0f && v[py[k][1]] > 0.0f) { gs += (s[py[k][1]] - s[py[k][0]]) * wy[k]; gw += wy[k]; }
        gy = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
        gs = gw = 0.0f;
        for (int k = 0; k < 4; k++)
          if (v[pz[k][0]] > 0.0f && v[pz[k][1]] > 0.0f) { gs += (s[pz[k][1]] - s[pz[k][0]]) * wz[k]; gw += wz[k]; }
        gz = gw > 0.0f ? gs / gw * inv_voxel : 0.0f;
      }
    }
  }
  g[0] = g[1] = g[2] = 0.0f;
  if (sdf >= max_dist) { *valid = 0; return max_dist; }
  const float nxg = -gx, nyg = -gy, nzg = -gz;
  const float len = sqrtf(nxg * nxg + nyg * nyg + nzg * nzg);
  if (len > 1e-6f) { g[0] = nxg / len; g[1] = nyg / len; g[2] = nzg / len; }
  return sdf;
}

/* ------------------------------------------------------------------------------------------
 * A5. uniform B-spline knots -> (pos, vel, acc, jerk) and its VJP
 *   kernels/trajectory/bspline/bspline_interpolation.cuh:95-297, bspline_boundary_constraint.cuh,
 *   basis/bspline_basis_matrix.cuh:20-156 (MATRIX backend, the one both reference backends
 *   select: cuda_core_backend/trajectory.py:71,163), bspline_context.cuh:73-170,
 *   bspline_gradient_util.cuh:141-227, bspline_common.cuh:15-47,138-179.
 * ---------------------------------------------------------------------------------------- */
static const float ORC_B3[4][4] = {{-1.0f / 6.0f, 3.0f / 6.0f, -3.0f / 6.0f, 1.0f / 6.0f},
                                   {3.0f / 6.0f, -6.0f / 6.0f, 0.0f, 4.0f / 6.0f},
                                   {-3.0f / 6.0f, 3.0f / 6.0f, 3.0f / 6.0f, 1.0f / 6.0f},
                                   {1.0f / 6.0f, 0.0f, 0.0f, 0.0f}};
static const float ORC_B4[5][5] = {
    {1.0f / 24.0f, -4.0f / 24.0f, 6.0f / 24.0f, -4.0f / 24.0f, 1.0f / 24.0f},
    {-4.0f / 24.0f, 12.0f / 24.0f, -6.0f / 24.0f, -12.0f / 24.0f, 11.0f / 24.0f},
    {6.0f / 24.0f, -12.0f / 24.0f, -6.0f / 24.0f, 12.0f / 24.0f, 11.0f / 24.0f},
    {-4.0f / 24.0f, 4.0f / 24.0f, 6.0f / 24.0f, 4.0f / 24.0f, 1.0f / 24.0f},
    {1.0f / 24.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
static const float ORC_B5[6][6] = {
    {-1.0f / 120.0f, 5.0f / 120.0f, -10.0f / 120.0f, 10.0f / 120.0f, -5.0f / 120.0f, 1.0f / 120.0f},
    {5.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 20.0f / 120.0f, -50.0f / 120.0f, 26.0f / 120.0f},
    {-10.0f / 120.0f, 30.0f / 120.0f, -0.0f / 120.0f, -60.0f / 120.0f, 0.0f / 120.0f, 66.0f / 120.0f},
    {10.0f / 120.0f, -20.0f / 120.0f, -20.0f / 120.0f, 20.0f / 120.0f, 50.0f / 120.0f, 26.0f / 120.0f},
    {-5.0f / 120.0f, 5.0f / 120.0f, 10.0f / 120.0f, 10.0f / 120.0f, 5.0f / 120.0f, 1.0f / 120.0f},
    {1.0f / 120.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};

static float orc_bcoef(int degree, int i, int j) {
  return degree == 3 ? ORC_B3[i][j] : (degree == 4 ? ORC_B4[i][j] : ORC_B5[i][j]);
}

/* basis for derivative order `der` (0..3): out[i] = sum_j COEF[i][j] * d^der/dt^der t^(deg-j)
 * (bspline_basis_matrix.cuh: compute_{position,velocity,acceleration,jerk}_basis; the partial
 * product uses the first (deg+1-der) columns). */
static void orc_bspline_basis(int degree, int der, float t, float *out) {
  const int n = degree + 1;
  float tp[6];
  const int m = n - der; /* number of monomials left */
  for (int j = 0; j < m; j++) {
    const int pw = degree - j; /* original power */
    /* falling factorial pw*(pw-1)*...*(pw-der+1) */
    float coef = 1.0f;
    for (int k = 0; k < der; k++) coef *= (float)(pw - k);
    float tv = 1.0f;
    for (int k = 0; k < pw - der; k++) tv *= t;
    tp[j] = coef * tv;
  }
  for (int i = 0; i < n; i++) {
    float acc = 0.0f;
    for (int j = 0; j < m; j++) acc += orc_bcoef(degree, i, j) * tp[j];
    out[i] = acc;
  }
}

/* bspline_boundary_constraint.cuh:52-92 */
static float orc_fixed_coef(int degree, int row, int col) {
  static const float c3[4][4] = {{1.0f, 1.0f, 1.0f, 1.0f},
                                 {-1.0f, 0.0f, 1.0f, 2.0f},
                                 {1.0f / 3.0f, -1.0f / 6.0f, 1.0f / 3.0f, 11.0f / 6.0f},
                                 {0.0f, 0.0f, 0.0f, 0.0f}};
  static const float c4[4][5] = {
      {1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
      {-3.0f / 2.0f, -1.0f / 2.0f, 1.0f / 2.0f, 3.0f / 2.0f, 5.0f / 2.0f},
      {11.0f / 12.0f, -1.0f / 12.0f, -1.0f / 12.0f, 11.0f / 12.0f, 35.0f / 12.0f},
      {-3.0f / 12.0f, 1.0f / 12.0f, -1.0f / 12.0f, 3.0f / 12.0f, 25.0f / 12.0f}};
  static const float c5[4][6] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                                 {-2.0f, -1.0f, 0.0f, 1.0f, 2.0f, 3.0f},
                                 {1.75f, 0.25f, -0.25f, 0.25f, 1.75f, 4.25f},
                                 {-0.833333f, 0.083333f, 0.0f, -0.083333f, 0.833333f, 3.75f}};
  return degree == 3 ? c3[row][col] : (degree == 4 ? c4[row][col] : c5[row][col]);
}

/* one (b, h, d) sample; bspline_interpolation.cuh:95-297 */
static void orc_bspline_sample(float *o4, const float *u_position, const float *start4[4],
                               const float *goal4[4], float interpolated_dt, int use_implicit_goal,
                               int padded_horizon, int dof, int b, int h, int d, int b_offset,
                               int goal_offset, int n_knots, int degree) {
  const int support = degree + 1;
  const int horizon = padded_horizon - 1;
  const int padded_n_knots = n_knots + support;
  const int interp = horizon / padded_n_knots;
  const float knot_dt = fmaxf(interpolated_dt, 1e-6f) * (float)interp; /* fp32Precision */
  float knots[6] = {0, 0, 0, 0, 0, 0};
  int knot_idx = interp > 0 ? h / interp : 0;
  int h_local = h;
  if (knot_idx >= padded_n_knots) { knot_idx = padded_n_knots - 1; h_local = -1; }
  const int start_knot_idx = knot_idx - support;
  for (int i = 0; i < support; i++) {
    const int src = start_knot_idx + i;
    if (src < n_knots && src >= 0) knots[i] = u_position[((size_t)b * n_knots + src) * dof + d];
  }
  const int req_start = knot_idx < support; /* bspline_common.cuh:24-27 */
  const int req_goal = use_implicit_goal ? (knot_idx > n_knots - 1) : (knot_idx > n_knots);
  float cpos = 0, cvel = 0, cacc = 0, cjerk = 0;
  if (req_start || req_goal) {
    const int ci = (req_start ? b_offset : goal_offset) * dof + d;
    const float **src = req_start ? start4 : goal4;
    cpos = src[0][ci]; cvel = src[1][ci]; cacc = src[2][ci]; cjerk = src[3][ci];
  }
  float t_mod = interp > 0 ? ((float)h / (float)interp) - (float)(int)(h / interp) : 0.0f;
  if (h_local < 0) t_mod = 1.0f;
  const float dt2 = knot_dt * knot_dt, dt3 = knot_dt * knot_dt * knot_dt;
  /* apply_boundary_constraints: bspline_boundary_constraint.cuh:330-367 */
  if (req_start || req_goal) {
    float fixed[6];
    for (int i = 0; i < support; i++)
      fixed[i] = orc_fixed_coef(degree, 0, i) * cpos + orc_fixed_coef(degree, 1, i) * cvel * knot_dt +
                 orc_fixed_coef(degree, 2, i) * cacc * dt2 + orc_fixed_coef(degree, 3, i) * cjerk * dt3;
    if (req_start) {
      const int loop = support - knot_idx;
      for (int i = 0; i < loop; i++) knots[i] = fixed[knot_idx + i];
    } else if (use_implicit_goal) {
      const int loop = knot_idx - n_knots + 1;
      const int st = support - loop;
      for (int i = 0; i < loop; i++) knots[st + i] = fixed[i];
    } else {
      const int loop = knot_idx - n_knots;
      const float v = knots[support - loop - 1];
      for (int i = 0; i < loop; i++) knots[support - i - 1] = v;
    }
  }
  float basis[6];
  const float scale[4] = {1.0f, knot_dt, dt2, dt3};
  for (int der = 0; der < 4; der++) {
    orc_bspline_basis(degree, der, t_mod, basis);
    float acc = 0.0f;
    for (int i = 0; i < support; i++) acc += knots[i] * basis[i];
    o4[der] = der == 0 ? acc : acc / scale[der];
  }
}

/* interpolate_bspline_kernel: bspline_kernel.cuh:81-151.  horizon here = padded horizon. */
ORC_API void orc_bspline_forward(float *out_pos, float *out_vel, float *out_acc, float *out_jerk,
                                 float *out_dt, const float *u_position, const float *start_pos,
                                 const float *start_vel, const float *start_acc,
                                 const float *start_jerk, const float *goal_pos,
                                 const float *goal_vel, const float *goal_acc,
                                 const float *goal_jerk, const int32_t *start_idx,
                                 const int32_t *goal_idx, const float *traj_dt,
                                 const uint8_t *use_implicit_goal_state, int batch,
                                 int padded_horizon, int dof, int n_knots, int degree) {
  const float *s4[4] = {start_pos, start_vel, start_acc, start_jerk};
  const float *g4[4] = {goal_pos, goal_vel, goal_acc, goal_jerk};
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const int bo = start_idx[b], go = goal_idx[b];
    const float dt = traj_dt[go];
    for (int h = 0; h < padded_horizon; h++)
      for (int d = 0; d < dof; d++) {
        float o[4];
        orc_bspline_sample(o, u_position, s4, g4, dt, use_implicit_goal_state[go], padded_horizon,
                           dof, b, h, d, bo, go, n_knots, degree);
        const size_t a = ((size_t)b * padded_horizon + h) * dof + d;
        out_pos[a] = o[0]; out_vel[a] = o[1]; out_acc[a] = o[2]; out_jerk[a] = o[3];
      }
    out_dt[b] = dt;
  }
}

/* interpolate_bspline_single_dt_kernel: bspline_kernel.cuh:221-270 (one dt, per-trajectory horizon,
 * output stride max_out; h runs over max_out and clamps to the last knot past the horizon) */
ORC_API void orc_bspline_single_dt(float *out_pos, float *out_vel, float *out_acc, float *out_jerk,
                                   float *out_dt, const float *knots, const float *start_pos,
                                   const float *start_vel, const float *start_acc,
                                   const float *start_jerk, const float *goal_pos,
                                   const float *goal_vel, const float *goal_acc,
                                   const float *goal_jerk, const int32_t *start_idx,
                                   const int32_t *goal_idx, const float *interpolation_dt,
                                   const uint8_t *use_implicit_goal_state,
                                   const int32_t *interpolation_horizon, int batch, int max_out,
                                   int dof, int n_knots, int degree) {
  const float *s4[4] = {start_pos, start_vel, start_acc, start_jerk};
  const float *g4[4] = {goal_pos, goal_vel, goal_acc, goal_jerk};
  const float dt = interpolation_dt[0];
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const int bo = start_idx[b], go = goal_idx[b];
    int nh = interpolation_horizon[b];
    if (nh > max_out - 1) nh = max_out - 1;
    for (int h = 0; h < max_out; h++)
      for (int d = 0; d < dof; d++) {
        float o[4];
        orc_bspline_sample(o, knots, s4, g4, dt, use_implicit_goal_state[go], nh + 1, dof, b, h, d, bo,
                           go, n_knots, degree);
        const size_t a = ((size_t)b * max_out + h) * dof + d;
        out_pos[a] = o[0]; out_vel[a] = o[1]; out_acc[a] = o[2]; out_jerk[a] = o[3];
      }
    out_dt[b] = dt;
  }
}

/* bspline_backward_kernel: bspline_kernel.cuh:332-380 with load_gradients
 * (bspline_gradient_util.cuh:141-227) and compute_backward_grad_from_basis
 * (bspline_context.cuh:133-170).  grad_* are [batch, padded_horizon, dof]. */
ORC_API void orc_bspline_backward(float *out_grad_knots, const float *grad_pos,
                                  const float *grad_vel, const float *grad_acc,
                                  const float *grad_jerk, const float *traj_dt,
                                  const int32_t *dt_idx, const uint8_t *use_implicit_goal_state,
                                  int batch, int padded_horizon, int dof, int n_knots, int degree) {
  const int support = degree + 1;
  const int horizon = padded_horizon - 1;
  const int total_knots = n_knots + support;
  const int interp = total_knots > 0 ? horizon / total_knots : 0;
  const int extended_horizon = total_knots * interp;
  const float *gin[4] = {grad_pos, grad_vel, grad_acc, grad_jerk};
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const int dto = dt_idx[b];
    const int use_goal = use_implicit_goal_state[dto];
    const float knot_dt = traj_dt[dto] * (float)interp; /* bspline_common.cuh:172 (no clamp) */
    const float scale[4] = {1.0f, knot_dt, knot_dt * knot_dt, knot_dt * knot_dt * knot_dt};
    for (int k = 0; k < n_knots; k++)
      for (int d = 0; d < dof; d++) {
        float total = 0.0f;
        float part[64]; /* the interpolation points' shares, summed below in the reference's order */
        for (int ii = 0; ii < interp; ii++) {
          float g[4][6];
          memset(g, 0, sizeof(g));
          const int implicit_goal_boundary = use_goal && k >= n_knots - 1;
          const int replicate_last = !use_goal && k == n_knots - 1;
          const size_t addr0 = (size_t)b * padded_horizon * dof + d;
          const int h_off = (k + 1) * interp + ii;
          for (int i = 0; i < support; i++) {
            const int hh = h_off + i * interp;
            if (hh < extended_horizon && !implicit_goal_boundary)
              for (int c = 0; c < 4; c++) g[c][i] = gin[c][addr0 + (size_t)hh * dof];
          }
          if (replicate_last) {
            for (int i = 1; i < support; i++)
              for (int x = 0; x < i; x++)
                for (int c = 0; c < 4; c++) g[c][x] += g[c][i];
            if (ii == 0) {
              const float tg = grad_pos[addr0 + (size_t)horizon * dof];
              for (int x = 0; x < support; x++) g[0][x] += tg;
            }
          }
          const int h_idx = (k + degree) * interp + ii;
          const float t_mod = ((float)h_idx / (float)interp) - (float)(int)(h_idx / interp);
          float basis[6];
          float sums[4];
          for (int der = 0; der < 4; der++) {
            orc_bspline_basis(degree, der, t_mod, basis);
            float acc = 0.0f;
            for (int i = 0; i < support; i++) acc += g[der][i] * basis[support - 1 - i];
            sums[der] = acc;
          }
          const float share = sums[0] + (sums[1] / scale[1]) + (sums[2] / scale[2]) + (sums[3] / scale[3]);
          if (ii < 64) part[ii] = share;
          total += share;
        }
        /* The reference adds the shares of a knot with a warp-segmented shuffle tree (bspline_gradient_util.cuh:83-105,
         * lane = interpolation index * knots_per_warp + knot): strides interp/2, interp/4, .., 1 -- e.g. (v0 + v2) + (v1 + v3)
         * for its default of four interpolation steps.  That tree is only a sum of the RIGHT lanes when the step count
         * divides the warp (a power of two): for 3, 5, 6, .. steps knots_per_warp * interp != 32 and the kernel pairs lanes of
         * different knots (run on the CPU it differs from the sum by O(1), tests/randomised/sweep_reference_kernels.py).  The
         * oracle follows the tree bit for bit where the reference is well defined and keeps the plain sum elsewhere. */
        if (interp >= 2 && interp <= 32 && (interp & (interp - 1)) == 0) {
          for (int stride = interp / 2; stride >= 1; stride /= 2)
            for (int i = 0; i < stride; i++) part[i] += part[i + stride];
          total = part[0];
        }
        out_grad_knots[((size_t)b * n_knots + k) * dof + d] = total;
      }
  }
}


/* ------------------------------------------------------------------------------------------
 * A5b. Legacy position-clique transition (differentiation_position_kernel.cuh) and acceleration
 *      integration (integration_acceleration_kernel.cuh).
 *
 * Forward (compute_central_difference :15-232, use_stencil = true as the launcher fixes it,
 * cuda_core_backend/trajectory.py:343): the branch table there is a 5-point window sliding over
 * ONE extended position sequence
 *   P = [e(-3), e(-2), e(-1), x0, u_0 .. u_{A-1}, u_{A-1} x4],   A = horizon - 4,
 * with e(.) the constant-acceleration back-extrapolation of the start state and u_{A-1} replaced
 * by the goal position when use_implicit_goal_state; window of point h = P[h .. h+4].
 * (identical to the reference's branches for horizon >= 9, where they do not overlap) */
static float orc_clique_P(int j, const float *u, int A, int dof, int d, float x0, float v0, float a0,
                          float dt, int use_goal, float goal) {
  const float fixed_jerk = 0.0f;
  if (j == 0) return (3.0f / 2) * (-1 * a0 * (dt * dt) - (dt * dt * dt) * fixed_jerk) - 3.0f * dt * v0 + x0;
  if (j == 1) return -2.0f * a0 * dt * dt - (4.0f / 3) * dt * dt * dt * fixed_jerk - 2.0f * dt * v0 + x0;
  if (j == 2) return -(3.0f / 2) * a0 * dt * dt - (7.0f / 6) * dt * dt * dt * fixed_jerk - dt * v0 + x0;
  if (j == 3) return x0;
  int i = j - 4;
  if (i >= A - 1) return use_goal ? goal : u[(size_t)(A - 1) * dof + d];
  return u[(size_t)i * dof + d];
}

ORC_API void orc_differentiation_position_forward(
    float *out_pos, float *out_vel, float *out_acc, float *out_jerk, float *out_dt, const float *u_position,
    const float *start_pos, const float *start_vel, const float *start_acc, const float *goal_pos,
    const int32_t *start_idx, const int32_t *goal_idx, const float *traj_dt,
    const uint8_t *use_implicit_goal_state, int batch, int horizon, int dof) {
  const int A = horizon - 4;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const int bo = start_idx[b], go = goal_idx[b];
    const float dt = traj_dt[go], dt_inv = 1.0f / dt;
    const int use_goal = use_implicit_goal_state[go];
    const float *u = u_position + (size_t)b * A * dof;
    for (int h = 0; h < horizon; h++)
      for (int d = 0; d < dof; d++) {
        const float x0 = start_pos[bo * dof + d], v0 = start_vel[bo * dof + d], a0 = start_acc[bo * dof + d];
        const float goal = use_goal ? goal_pos[go * dof + d] : 0.0f;
        float p[5];
        for (int i = 0; i < 5; i++) p[i] = orc_clique_P(h + i, u, A, dof, d, x0, v0, a0, dt, use_goal, goal);
        const size_t a = ((size_t)b * horizon + h) * dof + d;
        out_pos[a] = p[2];
        out_vel[a] = ((0.083333333f) * p[0] - (0.666666667f) * p[1] + (0.666666667f) * p[3] + (-0.083333333f) * p[4]) * dt_inv;
        out_acc[a] = ((-0.083333333f) * p[0] + (1.333333333f) * p[1] + (-2.5f) * p[2] + (1.333333333f) * p[3] +
                      (-0.083333333f) * p[4]) * dt_inv * dt_inv;
        out_jerk[a] = ((-(1.0f / 2.0f)) * p[0] + p[1] - p[3] + ((1.0f / 2.0f)) * p[4]) * (dt_inv * dt_inv * dt_inv);
      }
    out_dt[b] = dt;
  }
}

/* position_clique_loop_idx_bwd_kernel :234-370 (stencil variant) */
ORC_API void orc_differentiation_position_backward(
    float *out_grad, const float *grad_pos, const float *grad_vel, const float *grad_acc, const float *grad_jerk,
    const float *traj_dt, const int32_t *dt_idx, const uint8_t *use_implicit_goal_state, int batch, int horizon,
    int dof) {
  const int A = horizon - 4;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const int dto = dt_idx[b];
    const float dt_inv = 1.0f / traj_dt[dto];
    const int use_goal = use_implicit_goal_state[dto];
    const float i1 = dt_inv, i2 = dt_inv * dt_inv, i3 = dt_inv * dt_inv * dt_inv;
    for (int ah = 0; ah < A; ah++)
      for (int d = 0; d < dof; d++) {
        const size_t base = (size_t)b * horizon * dof + d;
        float gv[5], ga[5], gj[5];
        for (int i = 0; i < 5; i++) {
          gv[i] = grad_vel[base + (size_t)(ah + i) * dof];
          ga[i] = grad_acc[base + (size_t)(ah + i) * dof];
          gj[i] = grad_jerk[base + (size_t)(ah + i) * dof];
        }
        float g = grad_pos[base + (size_t)(ah + 2) * dof];
        if (ah == A - 1) {
          if (use_goal) g = 0.0f;
          else g += grad_pos[base + (size_t)(ah + 3) * dof] + grad_pos[base + (size_t)(ah + 4) * dof];
        }
        float o = g;
        if (ah < A - 1) {
          o += (float)((-0.083333333 * gv[0] + 0.666666667 * gv[1] - 0.666666667 * gv[3] + 0.083333333 * gv[4]) * i1);
          o += (float)((-0.083333333 * ga[0] + 1.333333333 * ga[1] + (-2.5) * ga[2] + 1.333333333 * ga[3] + (-0.083333333) * ga[4]) * i2);
          o += (0.5f * gj[0] - 1.0f * gj[1] + 1.0f * gj[3] - 0.5f * gj[4]) * i3;
        } else if (use_goal) {
          /* the last action is replaced by the goal in the forward pass: no gradient (differentiation_position_kernel.cuh:
           * 352-361 -- the stencil expression next to it there is commented out; found by running that kernel, oracle/_ref) */
          o = 0.0f;
        } else {
          o += (float)((-0.083333333 * gv[0] + 0.583333334 * gv[1] + 0.583333334 * gv[2] - 0.083333333 * gv[3]) * i1);
          o += (float)((-0.083333333 * ga[0] + 1.25 * ga[1] + (-1.25) * ga[2] + 0.083333333 * ga[3]) * i2);
          o += (float)((0.5 * gj[0] - 0.5 * gj[1] - 0.5 * gj[2] + 0.5 * gj[3]) * i3);
        }
        out_grad[((size_t)b * A + ah) * dof + d] = o;
      }
  }
}

/* acceleration_loop_idx(_rk2)_kernel: integration_acceleration_kernel.cuh:8-135 (both variants
 * run the same semi-implicit Euler recursion); traj_dt is indexed by the step h */
ORC_API void orc_integration_acceleration(float *out_pos, float *out_vel, float *out_acc, float *out_jerk,
                                          const float *u_acc, const float *start_pos, const float *start_vel,
                                          const float *start_acc, const int32_t *start_idx, const float *traj_dt,
                                          int batch, int horizon, int dof) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++)
    for (int d = 0; d < dof; d++) {
      const int bo = start_idx[b];
      float pos = start_pos[bo * dof + d], vel = start_vel[bo * dof + d], acc = start_acc[bo * dof + d];
      size_t a = (size_t)b * horizon * dof + d;
      out_pos[a] = pos; out_vel[a] = vel; out_acc[a] = acc; out_jerk[a] = 0.0f;
      for (int h = 1; h < horizon; h++) {
        const float dt = traj_dt[h];
        const float acc_n = u_acc[(size_t)b * horizon * dof + (size_t)(h - 1) * dof + d];
        vel = vel + acc_n * dt;
        pos = pos + vel * dt;
        a = ((size_t)b * horizon + h) * dof + d;
        out_acc[a] = acc_n; out_vel[a] = vel; out_pos[a] = pos; out_jerk[a] = (acc_n - acc) / dt;
        acc = acc_n;
      }
    }
}



/* c-space STATE cost (curobo/_src/cost/wp_cspace_state.py:20-287, helpers cost/warp_bound_util.py):
 * per (batch, horizon, dof): hinge^2 bound costs on position / velocity / acceleration / jerk /
 * effort (limits shrunk by activation_distance * range), optional joint-position target (scaled by
 * the non-terminal factor before the last step), squared-L2 regularisation of velocity /
 * acceleration / jerk / effort and the energy term (tau qd dt)^2.  weight, activation_distance and
 * squared_l2_regularization_weights have 5 entries; state_dt is per trajectory. */
static void orc_bound(float x, float lo, float hi, float w, float *cost, float *grad) {
  float delta;
  if (x < lo) delta = x - lo;
  else if (x > hi) delta = x - hi;
  else return;
  const float wv = w * delta;
  *cost += 0.5f * wv * delta;
  *grad += wv;
}
static void orc_sql2(float x, float w, float *cost, float *grad) {
  const float wv = w * x;
  *cost += 0.5f * wv * x;
  *grad += wv;
}
ORC_API void orc_cspace_state_cost(
    float *out_cost, float *out_gp, float *out_gv, float *out_ga, float *out_gj, float *out_gtau, const float *pos,
    const float *vel, const float *acc, const float *jerk, const float *effort, const float *state_dt,
    const float *target, const int32_t *idxs_target, const float *p_b, const float *v_b, const float *a_b,
    const float *j_b, const float *effort_b, const float *weight, const float *activation_distance,
    const float *sql2_weights, const float *target_weight, const float *non_terminal_factor,
    const float *target_dof_weight, int write_grad, int batch, int horizon, int dof, int retime_weights,
    int retime_regularization_weights) {
  const long total = (long)batch * horizon * dof;
#pragma omp parallel for schedule(static)
  for (long tid = 0; tid < total; tid++) {
    const int b = (int)(tid / ((long)horizon * dof));
    const int h = (int)((tid - (long)b * horizon * dof) / dof);
    const int d = (int)(tid % dof);
    const float dt = state_dt[b];
    float tw = target_weight[0];
    if (h < horizon - 1) tw *= non_terminal_factor[0];
    float wb[5], wr[5];
    for (int i = 0; i < 5; i++) { wb[i] = weight[i]; wr[i] = sql2_weights[i]; }
    if (retime_weights) { wb[1] = dt * wb[1]; wb[2] = powf(dt, 2.0f) * wb[2]; wb[3] = powf(dt, 3.0f) * wb[3]; }
    if (retime_regularization_weights) {
      wr[0] = dt * wr[0]; wr[1] = powf(dt, 2.0f) * wr[1]; wr[2] = powf(dt, 3.0f) * wr[2]; wr[4] = dt * wr[4];
    }
    const float x[5] = {pos[tid], vel[tid], acc[tid], jerk[tid], effort ? effort[tid] : 0.0f};
    const float *lim[5] = {p_b, v_b, a_b, j_b, effort_b};
    float c = 0.0f, g[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++) {
      float lo = lim[i][d], hi = lim[i][dof + d];
      const float r = hi - lo;
      lo = lo + activation_distance[i] * r;
      hi = hi - activation_distance[i] * r;
      orc_bound(x[i], lo, hi, wb[i], &c, &g[i]);
    }
    if (tw > 0.0f) {
      tw *= target_dof_weight[d];
      const float e = x[0] - target[(size_t)idxs_target[b] * dof + d];
      c += tw * e * e;
      g[0] += 2.0f * tw * e;
    }
    orc_sql2(x[1], wr[0], &c, &g[1]);
    orc_sql2(x[2], wr[1], &c, &g[2]);
    orc_sql2(x[3], wr[2], &c, &g[3]);
    orc_sql2(x[4], wr[3], &c, &g[4]);
    if (wr[4] > 0.0f) { /* aggregate_energy_regularization */
      const float e = x[4] * x[1] * dt;
      c += wr[4] * e * e;
      g[4] += 2.0f * wr[4] * e * x[1] * dt;
      g[1] += 2.0f * wr[4] * e * x[4] * dt;
    }
    out_cost[tid] = c;
    if (write_grad) { out_gp[tid] = g[0]; out_gv[tid] = g[1]; out_ga[tid] = g[2]; out_gj[tid] = g[3]; out_gtau[tid] = g[4]; }
  }
}

/* ------------------------------------------------------------------------------------------
 * A8. Inverse dynamics: body-frame RNEA and its VJP (config 4).
 *     kernels/dynamics/rnea_forward_kernel.cuh:53-292, rnea_backward_kernel.cuh:65-468,
 *     spatial_algebra.cuh, rnea_helpers.cuh; same algorithm as the reference's in-tree NumPy
 *     oracle curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py (which pins this file
 *     through tests/golden/rnea_golden.npz).
 *     Spatial vectors are [angular(3); linear(3)].  Links are visited in level_links order
 *     (parents before children); cache per (element, link) = v[6] a[6] f[6] 0 0
 *     (dynamics_constants.h:37-39).
 * ------------------------------------------------------------------------------------------ */
static void orc_local_Rp(const float *F, int jt, float q, float *R, float *p) {
  /* compute_local_Rp: R = R_fixed * R_joint(q), p = p_fixed (+ R_fixed * d for prismatic) */
  float Rf[9] = {F[0], F[1], F[2], F[4], F[5], F[6], F[8], F[9], F[10]};
  p[0] = F[3]; p[1] = F[7]; p[2] = F[11];
  for (int i = 0; i < 9; i++) R[i] = Rf[i];
  if (jt == J_FIXED) return;
  if (jt >= J_X_ROT) {
    const float c = cosf(q), s = sinf(q);
    const int ax = jt - J_X_ROT, a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
    /* rotation about axis ax: columns a1, a2 mix: col a1' = c col a1 + s col a2; col a2' = -s col a1 + c col a2 */
    for (int r = 0; r < 3; r++) {
      const float u = Rf[r * 3 + a1], w = Rf[r * 3 + a2];
      R[r * 3 + a1] = c * u + s * w;
      R[r * 3 + a2] = -s * u + c * w;
    }
  } else {
    const int ax = jt - J_X_PRISM;
    for (int r = 0; r < 3; r++) p[r] += Rf[r * 3 + ax] * q;
  }
}
static void orc_cross3(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
/* X v = [E w; E (v + w x p)], E = R^T */
static void orc_Xv(const float *R, const float *p, const float *v, float *o) {
  float t[3], c[3];
  orc_cross3(v, p, c);
  for (int i = 0; i < 3; i++) t[i] = v[3 + i] + c[i];
  for (int i = 0; i < 3; i++) {
    o[i] = R[0 * 3 + i] * v[0] + R[1 * 3 + i] * v[1] + R[2 * 3 + i] * v[2];
    o[3 + i] = R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2];
  }
}
/* X^T f = [R n + p x (R f); R f] */
static void orc_XTf(const float *R, const float *p, const float *f, float *o) {
  float Rf[3], Rn[3], c[3];
  for (int i = 0; i < 3; i++) {
    Rn[i] = R[i * 3 + 0] * f[0] + R[i * 3 + 1] * f[1] + R[i * 3 + 2] * f[2];
    Rf[i] = R[i * 3 + 0] * f[3] + R[i * 3 + 1] * f[4] + R[i * 3 + 2] * f[5];
  }
  orc_cross3(p, Rf, c);
  for (int i = 0; i < 3; i++) { o[i] = Rn[i] + c[i]; o[3 + i] = Rf[i]; }
}
/* I v with mc = [com xyz, mass], inertia = [ixx iyy izz ixy ixz iyz ..] at the CoM */
static void orc_Iv(const float *mc, const float *in, const float *v, float *o) {
  float h[3], c[3], ch[3];
  orc_cross3(v, mc, c);
  for (int i = 0; i < 3; i++) h[i] = v[3 + i] + c[i];
  orc_cross3(mc, h, ch);
  const float m = mc[3];
  o[0] = in[0] * v[0] + in[3] * v[1] + in[4] * v[2] + m * ch[0];
  o[1] = in[3] * v[0] + in[1] * v[1] + in[5] * v[2] + m * ch[1];
  o[2] = in[4] * v[0] + in[5] * v[1] + in[2] * v[2] + m * ch[2];
  for (int i = 0; i < 3; i++) o[3 + i] = m * h[i];
}
/* crf(v) f = [w x n + v x f; w x f] */
static void orc_crf(const float *v, const float *f, float *o) {
  float a[3], b[3], c[3];
  orc_cross3(v, f, a); orc_cross3(v + 3, f + 3, b); orc_cross3(v, f + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
/* crm(v1) v2 = [w1 x w2; v1 x w2 + w1 x v2] */
static void orc_crm(const float *v1, const float *v2, float *o) {
  float a[3], b[3], c[3];
  orc_cross3(v1, v2, a); orc_cross3(v1 + 3, v2, b); orc_cross3(v1, v2 + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static int orc_s_index(int jt) { return jt >= J_X_ROT ? jt - J_X_ROT : 3 + jt - J_X_PRISM; }

ORC_API void orc_rnea_forward(float *tau, float *forward_cache, const float *q, const float *qd, const float *qdd,
                              const float *fixed_transforms, const float *link_masses_com,
                              const float *link_inertias, const int8_t *joint_map_type, const int16_t *joint_map,
                              const int16_t *link_map, const float *joint_offset_map, const float *gravity,
                              const int16_t *level_links, const float *f_ext, int batch, int num_links, int num_dof) {
  const int L = num_links, D = num_dof;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    float *cache = forward_cache + (size_t)b * L * 20;
    float *v = (float *)malloc(sizeof(float) * L * 12), *a = v + L * 6;
    float *t = tau + (size_t)b * D;
    for (int j = 0; j < D; j++) t[j] = 0.0f;
    for (int idx = 0; idx < L; idx++) {
      const int k = level_links[idx];
      const int jt = joint_map_type[k], ji = joint_map[k], par = link_map[k];
      const int is_root = par < 0 || par == k;
      float qe = 0.f, qde = 0.f, qdde = 0.f;
      if (jt != J_FIXED && ji >= 0) {
        const float mul = joint_offset_map[2 * k];
        qe = mul * q[(size_t)b * D + ji] + joint_offset_map[2 * k + 1];
        qde = mul * qd[(size_t)b * D + ji];
        qdde = mul * qdd[(size_t)b * D + ji];
      }
      float R[9], p[3];
      orc_local_Rp(fixed_transforms + k * 12, jt, qe, R, p);
      float *vk = v + k * 6, *ak = a + k * 6;
      if (is_root) {
        for (int i = 0; i < 6; i++) vk[i] = 0.0f;
        orc_Xv(R, p, gravity, ak);
      } else {
        orc_Xv(R, p, v + par * 6, vk);
        orc_Xv(R, p, a + par * 6, ak);
      }
      if (jt != J_FIXED) {
        const int si = orc_s_index(jt);
        vk[si] += qde;
        ak[si] += qdde;
        float S[6] = {0, 0, 0, 0, 0, 0}, cor[6];
        S[si] = qde;
        orc_crm(vk, S, cor);
        for (int i = 0; i < 6; i++) ak[i] += cor[i];
      }
    }
    for (int k = 0; k < L; k++) {
      float *c = cache + k * 20;
      for (int i = 0; i < 6; i++) { c[i] = v[k * 6 + i]; c[6 + i] = a[k * 6 + i]; }
      float Ia[6], Ivv[6], x[6];
      orc_Iv(link_masses_com + k * 4, link_inertias + k * 8, a + k * 6, Ia);
      orc_Iv(link_masses_com + k * 4, link_inertias + k * 8, v + k * 6, Ivv);
      orc_crf(v + k * 6, Ivv, x);
      for (int i = 0; i < 6; i++)
        a[k * 6 + i] = Ia[i] + x[i] - (f_ext ? f_ext[((size_t)b * L + k) * 6 + i] : 0.0f); /* a now holds f */
    }
    for (int idx = L - 1; idx >= 0; idx--) {
      const int k = level_links[idx];
      const int jt = joint_map_type[k], ji = joint_map[k], par = link_map[k];
      const float *fk = a + k * 6;
      if (jt != J_FIXED && ji >= 0) t[ji] += joint_offset_map[2 * k] * fk[orc_s_index(jt)];
      if (!(par < 0 || par == k)) {
        float qe = 0.f;
        if (jt != J_FIXED && ji >= 0) qe = joint_offset_map[2 * k] * q[(size_t)b * D + ji] + joint_offset_map[2 * k + 1];
        float R[9], p[3], fc[6];
        orc_local_Rp(fixed_transforms + k * 12, jt, qe, R, p);
        orc_XTf(R, p, fk, fc);
        for (int i = 0; i < 6; i++) a[par * 6 + i] += fc[i];
      }
    }
    for (int k = 0; k < L; k++) {
      float *c = cache + k * 20 + 12;
      for (int i = 0; i < 6; i++) c[i] = a[k * 6 + i];
      c[6] = c[7] = 0.0f;
    }
    free(v);
  }
}

ORC_API void orc_rnea_backward(float *grad_q, float *grad_qd, float *grad_qdd, float *grad_f_ext,
                               const float *grad_tau, const float *q, const float *qd, const float *fixed_transforms,
                               const float *link_masses_com, const float *link_inertias,
                               const int8_t *joint_map_type, const int16_t *joint_map, const int16_t *link_map,
                               const float *joint_offset_map, const float *gravity, const int16_t *level_links,
                               const float *forward_cache, int batch, int num_links, int num_dof) {
  const int L = num_links, D = num_dof;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const float *cache = forward_cache + (size_t)b * L * 20;
    float *w = (float *)calloc((size_t)L * 18, sizeof(float));
    float *fb = w, *ab = w + L * 6, *vb = w + L * 12;
    float *gq = grad_q + (size_t)b * D, *gqd = grad_qd + (size_t)b * D, *gqdd = grad_qdd + (size_t)b * D;
    for (int j = 0; j < D; j++) gq[j] = gqd[j] = gqdd[j] = 0.0f;
    /* pass 1 (root -> leaves): adjoint of the force propagation */
    for (int idx = 0; idx < L; idx++) {
      const int k = level_links[idx];
      const int jt = joint_map_type[k], ji = joint_map[k], par = link_map[k];
      const int mov = jt != J_FIXED && ji >= 0;
      const float mul = mov ? joint_offset_map[2 * k] : 1.0f;
      float *fbk = fb + k * 6;
      if (mov) fbk[orc_s_index(jt)] += mul * grad_tau[(size_t)b * D + ji];
      if (!(par < 0 || par == k)) {
        const float qe = mov ? mul * q[(size_t)b * D + ji] + joint_offset_map[2 * k + 1] : 0.0f;
        float R[9], p[3], X[6];
        orc_local_Rp(fixed_transforms + k * 12, jt, qe, R, p);
        orc_Xv(R, p, fb + par * 6, X);
        for (int i = 0; i < 6; i++) fbk[i] += X[i];
        if (mov) { /* X_fbar_parent . crf(S) f_k */
          float S[6] = {0, 0, 0, 0, 0, 0}, cf[6];
          S[orc_s_index(jt)] = 1.0f;
          orc_crf(S, cache + k * 20 + 12, cf);
          float d = 0.0f;
          for (int i = 0; i < 6; i++) d += X[i] * cf[i];
          gq[ji] += mul * d;
        }
      }
    }
    if (grad_f_ext)
      for (int k = 0; k < L; k++)
        for (int i = 0; i < 6; i++) grad_f_ext[((size_t)b * L + k) * 6 + i] = -fb[k * 6 + i];
    /* pass 2 (leaves -> root): adjoint of the velocity / acceleration propagation */
    for (int idx = L - 1; idx >= 0; idx--) {
      const int k = level_links[idx];
      const int jt = joint_map_type[k], ji = joint_map[k], par = link_map[k];
      const int is_root = par < 0 || par == k;
      const int mov = jt != J_FIXED && ji >= 0;
      const float mul = mov ? joint_offset_map[2 * k] : 1.0f;
      const float *mc = link_masses_com + k * 4, *in = link_inertias + k * 8;
      const float *vk = cache + k * 20;
      float *abk = ab + k * 6, *vbk = vb + k * 6, *fbk = fb + k * 6;
      float t1[6], t2[6];
      orc_Iv(mc, in, fbk, t1);
      for (int i = 0; i < 6; i++) abk[i] += t1[i];
      orc_Iv(mc, in, vk, t1);
      orc_crf(fbk, t1, t2);
      for (int i = 0; i < 6; i++) vbk[i] -= t2[i];
      orc_crm(vk, fbk, t1);
      orc_Iv(mc, in, t1, t2);
      for (int i = 0; i < 6; i++) vbk[i] -= t2[i];
      const int si = mov ? orc_s_index(jt) : 0;
      if (mov) {
        const float qdk = mul * qd[(size_t)b * D + ji];
        gqdd[ji] += mul * abk[si];
        orc_crf(vk, abk, t1);
        gqd[ji] -= mul * t1[si];
        float S[6] = {0, 0, 0, 0, 0, 0};
        S[si] = qdk;
        orc_crf(S, abk, t1);
        for (int i = 0; i < 6; i++) vbk[i] += t1[i];
      }
      const float qe = mov ? mul * q[(size_t)b * D + ji] + joint_offset_map[2 * k + 1] : 0.0f;
      float R[9], p[3], S1[6] = {0, 0, 0, 0, 0, 0};
      orc_local_Rp(fixed_transforms + k * 12, jt, qe, R, p);
      S1[si] = 1.0f;
      if (!is_root) {
        orc_XTf(R, p, abk, t1);
        for (int i = 0; i < 6; i++) ab[par * 6 + i] += t1[i];
      }
      if (mov) { /* dX/dq on the acceleration path: parent acceleration, or gravity at the root */
        float Xa[6], cm[6];
        orc_Xv(R, p, is_root ? gravity : cache + par * 20 + 6, Xa);
        orc_crm(S1, Xa, cm);
        float d = 0.0f;
        for (int i = 0; i < 6; i++) d += abk[i] * cm[i];
        gq[ji] -= mul * d;
        gqd[ji] += mul * vbk[si];
      }
      if (!is_root) {
        orc_XTf(R, p, vbk, t1);
        for (int i = 0; i < 6; i++) vb[par * 6 + i] += t1[i];
        if (mov) {
          float Xv[6], cm[6];
          orc_Xv(R, p, cache + par * 20, Xv);
          orc_crm(S1, Xv, cm);
          float d = 0.0f;
          for (int i = 0; i < 6; i++) d += vbk[i] * cm[i];
          gq[ji] -= mul * d;
        }
      }
    }
    free(w);
  }
}


/* ------------------------------------------------------------------------------------------
 * A9. Levenberg-Marquardt step (optim/util/levenberg_marquardt_step.py:146-199, a Warp tile
 *     kernel: tile_matmul, tile_cholesky, tile_cholesky_solve).  fp32 throughout like the
 *     reference; pinned against numpy.linalg.solve in float64 (the Warp kernel cannot run).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_lm_step(float *q_out, float *pred, const float *jac, const float *jtr, const float *lam,
                         const float *q_in, int batch, int n_res, int dof) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    const float *J = jac + (size_t)b * n_res * dof;
    float *A = (float *)malloc(sizeof(float) * dof * dof), *y = (float *)malloc(sizeof(float) * dof);
    for (int i = 0; i < dof; i++)
      for (int j = 0; j < dof; j++) {
        float s = 0.0f;
        for (int k = 0; k < n_res; k++) s += J[k * dof + i] * J[k * dof + j];
        A[i * dof + j] = s + (i == j ? lam[b] : 0.0f);
      }
    for (int j = 0; j < dof; j++) { /* Cholesky, lower triangle in place */
      for (int i = j; i < dof; i++) {
        float s = A[i * dof + j];
        for (int k = 0; k < j; k++) s -= A[i * dof + k] * A[j * dof + k];
        A[i * dof + j] = (i == j) ? sqrtf(s) : s / A[j * dof + j];
      }
    }
    for (int i = 0; i < dof; i++) {
      float s = -jtr[(size_t)b * dof + i];
      for (int k = 0; k < i; k++) s -= A[i * dof + k] * y[k];
      y[i] = s / A[i * dof + i];
    }
    for (int i = dof - 1; i >= 0; i--) {
      float s = y[i];
      for (int k = i + 1; k < dof; k++) s -= A[k * dof + i] * y[k];
      y[i] = s / A[i * dof + i];
    }
    float red = 0.0f;
    for (int i = 0; i < dof; i++) {
      q_out[(size_t)b * dof + i] = q_in[(size_t)b * dof + i] + y[i];
      red += y[i] * (lam[b] * y[i] - jtr[(size_t)b * dof + i]);
    }
    pred[b] = 0.5f * red;
    free(A); free(y);
  }
}

/* ------------------------------------------------------------------------------------------
 * A6. L-BFGS step (lbfgs_step_kernel.cuh:18-90, lbfgs_step_helpers.cuh:37-470)
 *   Buffers: y_buffer, s_buffer [m, B, V]; rho_buffer [m, B]; x_0, grad_0, q, grad_q, step [B, V].
 *   stable_mode follows the CUDA kernel (rho = 0 when y.s <= 0), which differs from the torch
 *   twin (nan_to_num only) exactly when y.s < 0.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_lbfgs_step(float *step_vec, float *rho_buffer, float *y_buffer, float *s_buffer,
                            const float *q, const float *grad_q, float *x_0, float *grad_0,
                            float epsilon, int batch, int m, int v_dim, int stable_mode) {
#pragma omp parallel
  {
    float *gq = (float *)malloc((size_t)v_dim * sizeof(float));
    float *alpha = (float *)malloc((size_t)(m > 0 ? m : 1) * sizeof(float));
#pragma omp for schedule(static)
    for (int b = 0; b < batch; b++) {
      const size_t bv = (size_t)b * v_dim;
      float numerator = 0.0f;
      /* load + differences (:37-70); history shift + append (:89-150) */
      for (int i = 0; i < m - 1; i++)
        for (int v = 0; v < v_dim; v++) {
          y_buffer[((size_t)i * batch) * v_dim + bv + v] = y_buffer[((size_t)(i + 1) * batch) * v_dim + bv + v];
          s_buffer[((size_t)i * batch) * v_dim + bv + v] = s_buffer[((size_t)(i + 1) * batch) * v_dim + bv + v];
        }
      for (int v = 0; v < v_dim; v++) {
        const float g = grad_q[bv + v], x = q[bv + v];
        const float y = g - grad_0[bv + v], s = x - x_0[bv + v];
        grad_0[bv + v] = g; x_0[bv + v] = x;
        gq[v] = g;
        numerator += y * s;
        if (m > 0) {
          y_buffer[((size_t)(m - 1) * batch) * v_dim + bv + v] = y;
          s_buffer[((size_t)(m - 1) * batch) * v_dim + bv + v] = s;
        }
      }
      /* update_rho_buffer (:213-240) */
      for (int i = 0; i < m - 1; i++) rho_buffer[(size_t)i * batch + b] = rho_buffer[(size_t)(i + 1) * batch + b];
      if (m > 0) {
        float rho = 1.0f / numerator;
        if (stable_mode && numerator <= 0.0f) rho = 0.0f;
        rho_buffer[(size_t)(m - 1) * batch + b] = rho;
      }
      /* backward pass (:262-330) */
      for (int i = m - 1; i >= 0; i--) {
        const float *si = s_buffer + ((size_t)i * batch) * v_dim + bv;
        const float *yi = y_buffer + ((size_t)i * batch) * v_dim + bv;
        float dot = 0.0f;
        for (int v = 0; v < v_dim; v++) dot += gq[v] * si[v];
        alpha[i] = dot * rho_buffer[(size_t)i * batch + b];
        for (int v = 0; v < v_dim; v++) gq[v] = gq[v] - alpha[i] * yi[v];
      }
      /* scaling (:346-373) */
      if (m > 0) {
        const float *yl = y_buffer + ((size_t)(m - 1) * batch) * v_dim + bv;
        float den = 0.0f;
        for (int v = 0; v < v_dim; v++) den += yl[v] * yl[v];
        float var1 = numerator / den;
        if (stable_mode && (isinf(var1) || isnan(var1))) var1 = epsilon;
        const float gamma = var1 < 0 ? 0 : var1;
        for (int v = 0; v < v_dim; v++) gq[v] = gamma * gq[v];
      }
      /* forward pass (:396-470) */
      for (int i = 0; i < m; i++) {
        const float *si = s_buffer + ((size_t)i * batch) * v_dim + bv;
        const float *yi = y_buffer + ((size_t)i * batch) * v_dim + bv;
        float dot = 0.0f;
        for (int v = 0; v < v_dim; v++) dot += gq[v] * yi[v];
        const float beta = alpha[i] - dot * rho_buffer[(size_t)i * batch + b];
        for (int v = 0; v < v_dim; v++) gq[v] = gq[v] + beta * si[v];
      }
      for (int v = 0; v < v_dim; v++) step_vec[bv + v] = -gq[v];
    }
    free(gq);
    free(alpha);
  }
}

/* ------------------------------------------------------------------------------------------
 * A7. Wolfe line search + best/convergence bookkeeping
 *   line_search_kernel.cuh:27-155, line_search_helpers.cuh:18-316.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_line_search(float *best_cost, float *best_action, int16_t *best_iteration,
                             int16_t *current_iteration, uint8_t *converged,
                             int convergence_iteration, float cost_delta_threshold,
                             float cost_relative_threshold, float *exploration_cost,
                             float *exploration_action, float *exploration_gradient,
                             int32_t *exploration_idx, float *selected_cost,
                             float *selected_action, float *selected_gradient,
                             int32_t *selected_idx, const float *search_cost,
                             const float *search_action, const float *search_gradient,
                             const float *step_direction, const float *search_magnitudes,
                             float c_1, float c_2, int strong_wolfe, int approx_wolfe,
                             int n_linesearch, int opt_dim, int batch) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    float gd[64];
    const float *dir = step_direction + (size_t)b * opt_dim;
    for (int i = 0; i < n_linesearch; i++) {
      const float *g = search_gradient + ((size_t)b * n_linesearch + i) * opt_dim;
      float acc = 0.0f;
      for (int v = 0; v < opt_dim; v++) acc += g[v] * dir[v];
      gd[i] = acc;
    }
    /* evaluate_wolfe_conditions (:262-312) + compute_wolfe_indices (:62-95) */
    int id1 = 0, id = 0; /* largest index with armijo / with both, 0 if none */
    const float c0 = search_cost[(size_t)b * n_linesearch];
    for (int i = 0; i < n_linesearch; i++) {
      const float a = search_magnitudes[i];
      const float cv = search_cost[(size_t)b * n_linesearch + i];
      const int w1 = cv <= (c0 + c_1 * a * gd[0]);
      const int w2 = strong_wolfe ? (fabsf(gd[i]) <= c_2 * fabsf(gd[0])) : (gd[i] >= c_2 * gd[0]);
      if (w1) id1 = i;
      if (w1 && w2) id = i;
    }
    /* get_linesearch_idx (:46-60) */
    const int sel = strong_wolfe ? id : (id == 0 ? id1 : id);
    const int expl = (approx_wolfe && !strong_wolfe && sel == 0) ? 1 : sel;
    /* update_costs_and_convergence (:97-150) + check_best_convergence (:18-44) */
    exploration_cost[b] = search_cost[(size_t)b * n_linesearch + expl];
    const float sc = search_cost[(size_t)b * n_linesearch + sel];
    selected_cost[b] = sc;
    const float bc = best_cost[b];
    const int cur = current_iteration[b] + 1;
    int bi = best_iteration[b];
    const float delta = bc - sc;
    const float rel = delta / (bc + 1e-6f);
    const int update_best = delta > cost_delta_threshold && rel > cost_relative_threshold;
    if (update_best) bi = cur;
    converged[b] = (bi + convergence_iteration < cur) ? 1 : 0;
    best_iteration[b] = (int16_t)bi;
    current_iteration[b] = (int16_t)cur;
    if (update_best) best_cost[b] = sc;
    /* copy_action_gradient_results (:152-198) */
    for (int v = 0; v < opt_dim; v++) {
      const size_t es = ((size_t)b * n_linesearch + expl) * opt_dim + v;
      const size_t ss = ((size_t)b * n_linesearch + sel) * opt_dim + v;
      exploration_action[(size_t)b * opt_dim + v] = search_action[es];
      exploration_gradient[(size_t)b * opt_dim + v] = search_gradient[es];
      selected_action[(size_t)b * opt_dim + v] = search_action[ss];
      selected_gradient[(size_t)b * opt_dim + v] = search_gradient[ss];
      if (update_best) best_action[(size_t)b * opt_dim + v] = search_action[ss];
    }
    for (int i = 0; i < n_linesearch; i++) {
      exploration_idx[(size_t)b * n_linesearch + i] = expl;
      selected_idx[(size_t)b * n_linesearch + i] = sel;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * A8. per-trajectory cost sum (rollout/metrics.py:233-265, util/tensor_util.py:104 cat_sum):
 *     out[b] = sum_h ( self[b,h] + sum_s scene[b,h,s] ).  Used by the rollout parity tests.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_trajectory_cost_sum(float *out, const float *self_cost, const float *scene_cost,
                                     int batch, int horizon, int nspheres) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; b++) {
    double acc = 0.0;
    for (int h = 0; h < horizon; h++) {
      if (self_cost) acc += self_cost[(size_t)b * horizon + h];
      if (scene_cost)
        for (int s = 0; s < nspheres; s++) acc += scene_cost[((size_t)b * horizon + h) * nspheres + s];
    }
    out[b] = (float)acc;
  }
}

/* ------------------------------------------------------------------------------------------
 * A9. goal-set tool-pose distance (Warp in the reference): cost/wp_tool_pose.py:61-435 (error
 *     functions), :456-692 (kernel).  Quaternions are stored wxyz in memory and handled xyzw
 *     inside, as in the reference.  rotation_method: 0 axis-angle, 1 lie group, 2 lie advanced.
 *     The rotation gradient is emitted as a quaternion RATE  q (x) (omega, 0)  (wxyz), which the
 *     FK backward turns back into omega/2 (quaternion_util.cuh:86-102).
 * ---------------------------------------------------------------------------------------- */
static void orc_qmul(const float *a, const float *b, float *o) { /* xyzw Hamilton product */
  const float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  o[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
  o[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
  o[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
  o[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
}

static void orc_rotation_error(const float *cq, const float *gq, const float *wt, float rw, float tol,
                               int method, float *dist, float *grad_w, float *angle_out) {
  const float ginv[4] = {-gq[0], -gq[1], -gq[2], gq[3]};
  float qd[4];
  orc_qmul(cq, ginv, qd);
  grad_w[0] = grad_w[1] = grad_w[2] = 0.0f;
  if (method == 0) { /* wp_tool_pose.py:129-203 */
    const float v[3] = {wt[0] * qd[0], wt[1] * qd[1], wt[2] * qd[2]};
    const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    float angle = 2.0f * atan2f(len, fabsf(qd[3]));
    if (rw == 0.0f) angle = 0.0f;
    float ax[3] = {0, 0, 0};
    if (!(len < 1e-15f)) { ax[0] = v[0] / len; ax[1] = v[1] / len; ax[2] = v[2] / len; }
    const float om[3] = {angle * ax[0], angle * ax[1], angle * ax[2]};
    float d = rw * (om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    if (d < tol) d = 0.0f;
    else {
      float sf = 2.0f;
      if (qd[3] < 0.0f) sf = -1.0f * sf;
      for (int k = 0; k < 3; k++) grad_w[k] = sf * rw * om[k];
    }
    *dist = d; *angle_out = angle;
    return;
  }
  /* lie group (1) and lie group advanced (2) produce identical outputs: :206-383 */
  if (qd[3] < 0.0f) { qd[0] = -qd[0]; qd[1] = -qd[1]; qd[2] = -qd[2]; qd[3] = -qd[3]; }
  const float w = qd[3];
  const float vn = sqrtf(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2]);
  float half = atan2f(vn, fabsf(w));
  if (rw == 0.0f) half = 0.0f;
  const float geo = 2.0f * half;
  float tv[3];
  if (vn < 1e-10f) { for (int k = 0; k < 3; k++) tv[k] = 2.0f * qd[k]; }
  else if (fabsf(half) < 1e-15f) {
    const float corr = 1.0f + (vn * vn) / (6.0f * w * w);
    for (int k = 0; k < 3; k++) tv[k] = 2.0f * qd[k] * corr;
  } else {
    const float sinc = geo / (2.0f * sinf(half));
    for (int k = 0; k < 3; k++) tv[k] = sinc * qd[k];
  }
  const float wv[3] = {wt[0] * tv[0], wt[1] * tv[1], wt[2] * tv[2]};
  const float n2 = wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2];
  float d = rw * n2;
  if (d < tol) d = 0.0f;
  else for (int k = 0; k < 3; k++) grad_w[k] = 2.0f * rw * wv[k];
  *dist = d; *angle_out = sqrtf(n2);
}

ORC_API void orc_tool_pose_distance(
    float *out_distance, float *out_position_distance, float *out_rotation_distance,
    float *out_position_gradient, float *out_rotation_gradient, int32_t *out_goalset_idx,
    const float *current_position, const float *current_quat, const float *goal_position,
    const float *goal_quat, const int32_t *idxs_goal, const float *position_orientation_weight,
    const float *terminal_axes_weight, const float *non_terminal_axes_weight,
    const float *terminal_tolerance, const float *non_terminal_tolerance,
    const uint8_t *project_distance_to_goal, int batch, int horizon, int num_links, int num_goalset,
    int rotation_method) {
  const long total = (long)batch * horizon * num_links;
#pragma omp parallel for schedule(static)
  for (long tid = 0; tid < total; tid++) {
    const int b = (int)(tid / ((long)horizon * num_links));
    const int h = (int)((tid - (long)b * horizon * num_links) / num_links);
    const int l = (int)(tid - (long)b * horizon * num_links - (long)h * num_links);
    const int non_terminal = (h < horizon - 1) && horizon > 1;
    const float *aw = (non_terminal ? non_terminal_axes_weight : terminal_axes_weight) + l * 6;
    const float *tl = (non_terminal ? non_terminal_tolerance : terminal_tolerance) + l * 2;
    const float pw = position_orientation_weight[0], rw = position_orientation_weight[1];
    const float tol_p = tl[0] * tl[0], tol_r = tl[1] * tl[1];
    const int gi = idxs_goal[b];
    const int project = project_distance_to_goal[l];
    const float *cp = current_position + tid * 3;
    const float *cqw = current_quat + tid * 4;
    const float cq[4] = {cqw[1], cqw[2], cqw[3], cqw[0]};
    float best = -1.0f, best_pd = -1.0f, best_rd = -1.0f, best_angle = -1.0f;
    float best_pg[3] = {0, 0, 0}, best_rg[3] = {0, 0, 0}, best_gq[4] = {0, 0, 0, 1};
    int best_g = 0;
    for (int g = 0; g < num_goalset; g++) {
      const size_t ga = ((size_t)gi * num_links + l) * num_goalset + g;
      const float *gp = goal_position + ga * 3;
      const float *gqw = goal_quat + ga * 4;
      const float gq[4] = {gqw[1], gqw[2], gqw[3], gqw[0]};
      float cpf[3], cqf[4], gpf[3], gqf[4];
      if (project == 1) { /* current pose expressed in the goal frame, goal = identity */
        const float gi_q[4] = {-gq[0], -gq[1], -gq[2], gq[3]};
        const float dpv[3] = {cp[0] - gp[0], cp[1] - gp[1], cp[2] - gp[2]};
        orc_quat_rotate(gi_q, dpv, cpf);
        orc_qmul(gi_q, cq, cqf);
        gpf[0] = gpf[1] = gpf[2] = 0.0f;
        gqf[0] = gqf[1] = gqf[2] = 0.0f; gqf[3] = 1.0f;
      } else {
        memcpy(cpf, cp, 12); memcpy(cqf, cq, 16); memcpy(gpf, gp, 12); memcpy(gqf, gq, 16);
      }
      /* compute_position_error: :61-104 */
      const float dl[3] = {cpf[0] - gpf[0], cpf[1] - gpf[1], cpf[2] - gpf[2]};
      const float wd[3] = {dl[0] * aw[0], dl[1] * aw[1], dl[2] * aw[2]};
      float pd = 0.5f * pw * (wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
      float pg[3] = {pw * aw[0] * aw[0] * dl[0], pw * aw[1] * aw[1] * dl[1], pw * aw[2] * aw[2] * dl[2]};
      if (pd < tol_p) { pd = 0.0f; pg[0] = pg[1] = pg[2] = 0.0f; }
      float rd, rg[3], angle;
      orc_rotation_error(cqf, gqf, aw + 3, rw, tol_r, rotation_method, &rd, rg, &angle);
      const float tot = pd + rd;
      if (best < 0 || tot < best) {
        best = tot; best_g = g; best_pd = pd; best_rd = rd; best_angle = angle;
        memcpy(best_pg, pg, 12); memcpy(best_rg, rg, 12); memcpy(best_gq, gq, 16);
      }
    }
    if (project == 1) { /* gradients back to the world frame */
      float t3[3];
      orc_quat_rotate(best_gq, best_pg, t3); memcpy(best_pg, t3, 12);
      orc_quat_rotate(best_gq, best_rg, t3); memcpy(best_rg, t3, 12);
    }
    const float om[4] = {best_rg[0], best_rg[1], best_rg[2], 0.0f};
    float qr[4];
    orc_qmul(cq, om, qr); /* convert_angular_velocity_to_quaternion_rate: :107-126 */
    out_distance[2 * tid] = best_pd;
    out_distance[2 * tid + 1] = best_rd;
    out_goalset_idx[tid] = best_g;
    out_position_distance[tid] = pw > 0.0f ? sqrtf(2.0f * best_pd / pw) : 0.0f;
    out_rotation_distance[tid] = best_angle;
    memcpy(out_position_gradient + tid * 3, best_pg, 12);
    out_rotation_gradient[tid * 4 + 0] = qr[3];
    out_rotation_gradient[tid * 4 + 1] = qr[0];
    out_rotation_gradient[tid * 4 + 2] = qr[1];
    out_rotation_gradient[tid * 4 + 3] = qr[2];
  }
}

/* ------------------------------------------------------------------------------------------
 * A10. c-space POSITION cost (Warp): cost/wp_cspace_position.py:232-362, warp_bound_util.py:9-100.
 *   Joint-limit (and optional effort-limit) hinge^2, optional target and implied
 *   velocity/acceleration regularisation w.r.t. a current state.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_cspace_position_cost(
    float *out_cost, float *out_grad_p, float *out_grad_tau, const float *pos, const float *effort,
    const float *cspace_target, const int32_t *cspace_target_idx, const float *p_b,
    const float *effort_b, const float *weight, const float *activation_distance,
    const float *cspace_target_weight, const float *cspace_target_dof_weight,
    const float *squared_l2_reg_weight, const float *current_position,
    const float *current_velocity, const int32_t *idxs_current_state, const float *v_b,
    const float *state_dt, int write_grad, int batch, int horizon, int dof) {
  const long total = (long)batch * horizon * dof;
#pragma omp parallel for schedule(static)
  for (long tid = 0; tid < total; tid++) {
    const int b = (int)(tid / ((long)horizon * dof));
    const int d = (int)(tid % dof);
    const float eta_p = activation_distance[0], eta_tau = activation_distance[1];
    const float w = weight[0], tau_w = weight[1];
    float tl = effort_b[d], tu = effort_b[dof + d];
    { const float r = tu - tl; tl = tl + eta_tau * r; tu = tu - eta_tau * r; }
    float pl = p_b[d], pu = p_b[dof + d];
    { const float r = pu - pl; pl = pl + eta_p * r; pu = pu - eta_p * r; }
    const int cur = idxs_current_state[b];
    const float dt = state_dt[cur];
    float cur_p = 0.0f;
    if (dt > 0.0f) {
      cur_p = current_position[(size_t)cur * dof + d];
      pl = fmaxf(pl, cur_p + v_b[d] * dt);
      pu = fminf(pu, cur_p + v_b[dof + d] * dt);
    }
    const float cp = pos[tid];
    const float ctau = effort ? effort[tid] : 0.0f;
    float c = 0.0f, gp = 0.0f, gt = 0.0f;
    if (cp < pl || cp > pu) {
      const float delta = cp < pl ? cp - pl : cp - pu;
      const float wv = w * delta;
      c += 0.5f * wv * delta; gp += wv;
    }
    if (tau_w > 0.0f && (ctau < tl || ctau > tu)) {
      const float delta = ctau < tl ? ctau - tl : ctau - tu;
      const float wv = tau_w * delta;
      c += 0.5f * wv * delta; gt += wv;
    }
    const float tw = cspace_target_weight[0] * cspace_target_dof_weight[d];
    if (tw > 0.0f) {
      const float e = cp - cspace_target[(size_t)cspace_target_idx[b] * dof + d];
      c += tw * e * e; gp += 2.0f * tw * e;
    }
    const float vw = squared_l2_reg_weight[0] * dt, aw = squared_l2_reg_weight[1] * dt * dt;
    if (dt > 0.0f && (vw > 0.0f || aw > 0.0f)) {
      const float vi = (cp - cur_p) / dt;
      if (vw > 0.0f) { c += 0.5f * vw * vi * vi; gp += vw * vi / dt; }
      if (aw > 0.0f) {
        const float ai = (vi - current_velocity[(size_t)cur * dof + d]) / dt;
        c += 0.5f * aw * ai * ai; gp += aw * ai / (dt * dt);
      }
    }
    out_cost[tid] = c;
    if (write_grad) { out_grad_p[tid] = gp; out_grad_tau[tid] = gt; }
  }
}
