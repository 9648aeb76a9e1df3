// cost_device.hpp -- device functions of the tool-pose and c-space costs, shared by cost.hip and
// the fused rollout kernels.  Reference: cost/wp_tool_pose.py:61-692, cost/wp_cspace_position.py
// :232-362, cost/warp_bound_util.py:9-100.
#pragma once
#include "common.hpp"

namespace curobo_hip {

struct ToolPoseArgs {
  float *out_distance, *out_position_distance, *out_rotation_distance, *out_position_gradient;
  float *out_rotation_gradient;
  int32_t *out_goalset_idx;
  const float *current_position, *current_quat, *goal_position, *goal_quat;
  const int32_t *idxs_goal;
  const float *position_orientation_weight, *terminal_axes_weight, *non_terminal_axes_weight;
  const float *terminal_tolerance, *non_terminal_tolerance;
  const uint8_t *project_distance_to_goal;
  int batch, horizon, num_links, num_goalset, rotation_method;
};

struct ToolPoseResult {
  float position_cost, rotation_cost, position_distance, rotation_distance;
  f3 position_gradient;
  float4 quat_rate_wxyz;
  int goalset_idx;
};

struct Quat {  // x, y, z, w
  float x, y, z, w;
};
__device__ __forceinline__ Quat qmul(Quat a, Quat b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ f3 qrot(Quat q, f3 v) {  // warp-lang quat_rotate
  const f3 qv = make_f3(q.x, q.y, q.z);
  const f3 c = cross(qv, v);
  const float d = dot(qv, v), k = 2.0f * q.w * q.w - 1.0f;
  return make_f3(v.x * k + c.x * q.w * 2.0f + q.x * d * 2.0f, v.y * k + c.y * q.w * 2.0f + q.y * d * 2.0f,
                 v.z * k + c.z * q.w * 2.0f + q.z * d * 2.0f);
}

// wp_tool_pose.py:129-383
__device__ __forceinline__ void rotation_error(Quat cq, Quat gq, f3 wt, float rw, float tol, int method, float &dist,
                                               f3 &grad_w, float &angle_out) {
  Quat qd = qmul(cq, Quat{-gq.x, -gq.y, -gq.z, gq.w});
  grad_w = make_f3(0.f, 0.f, 0.f);
  if (method == 0) {
    const f3 v = make_f3(wt.x * qd.x, wt.y * qd.y, wt.z * qd.z);
    const float len = sqrtf(dot(v, v));
    float angle = 2.0f * atan2f(len, fabsf(qd.w));
    if (rw == 0.0f) angle = 0.0f;
    f3 ax = make_f3(0.f, 0.f, 0.f);
    if (!(len < 1e-15f)) ax = make_f3(v.x / len, v.y / len, v.z / len);
    const f3 om = angle * ax;
    float d = rw * dot(om, om);
    if (d < tol) d = 0.0f;
    else {
      float sf = 2.0f;
      if (qd.w < 0.0f) sf = -1.0f * sf;
      grad_w = (sf * rw) * om;
    }
    dist = d;
    angle_out = angle;
    return;
  }
  if (qd.w < 0.0f) qd = Quat{-qd.x, -qd.y, -qd.z, -qd.w};
  const float w = qd.w;
  const f3 v = make_f3(qd.x, qd.y, qd.z);
  const float vn = sqrtf(dot(v, v));
  float half = atan2f(vn, fabsf(w));
  if (rw == 0.0f) half = 0.0f;
  const float geo = 2.0f * half;
  f3 tv;
  if (vn < 1e-10f) tv = 2.0f * v;
  else if (fabsf(half) < 1e-15f) tv = (2.0f * (1.0f + (vn * vn) / (6.0f * w * w))) * v;
  else tv = (geo / (2.0f * sinf(half))) * v;
  const f3 wv = make_f3(wt.x * tv.x, wt.y * tv.y, wt.z * tv.z);
  const float n2 = dot(wv, wv);
  float d = rw * n2;
  if (d < tol) d = 0.0f;
  else grad_w = (2.0f * rw) * wv;
  dist = d;
  angle_out = sqrtf(n2);
}

// one (batch, horizon, link) entry of the goal-set kernel, wp_tool_pose.py:456-692
__device__ __forceinline__ ToolPoseResult tool_pose_distance_point(const ToolPoseArgs &a, int b, int h, int l, f3 cp,
                                                                   float4 cq_wxyz) {
  const bool non_terminal = (h < a.horizon - 1) && a.horizon > 1;
  const float *aw = (non_terminal ? a.non_terminal_axes_weight : a.terminal_axes_weight) + l * 6;
  const float *tl = (non_terminal ? a.non_terminal_tolerance : a.terminal_tolerance) + l * 2;
  const float pw = a.position_orientation_weight[0], rw = a.position_orientation_weight[1];
  const float tol_p = tl[0] * tl[0], tol_r = tl[1] * tl[1];
  const f3 wpos = make_f3(aw[0], aw[1], aw[2]), wrot = make_f3(aw[3], aw[4], aw[5]);
  const int gi = a.idxs_goal[b];
  const bool project = a.project_distance_to_goal[l] == 1;
  const Quat cq{cq_wxyz.y, cq_wxyz.z, cq_wxyz.w, cq_wxyz.x};
  float best = -1.0f, best_pd = -1.0f, best_rd = -1.0f, best_angle = -1.0f;
  f3 best_pg = make_f3(0.f, 0.f, 0.f), best_rg = make_f3(0.f, 0.f, 0.f);
  Quat best_gq{0.f, 0.f, 0.f, 1.f};
  int best_g = 0;
  for (int g = 0; g < a.num_goalset; g++) {
    const size_t ga = ((size_t)gi * a.num_links + l) * a.num_goalset + g;
    const float *gp3 = a.goal_position + ga * 3;
    const float4 gqw = reinterpret_cast<const float4 *>(a.goal_quat)[ga];
    const f3 gp = make_f3(gp3[0], gp3[1], gp3[2]);
    const Quat gq{gqw.y, gqw.z, gqw.w, gqw.x};
    f3 cpf = cp, gpf = gp;
    Quat cqf = cq, gqf = gq;
    if (project) {
      const Quat gi_q{-gq.x, -gq.y, -gq.z, gq.w};
      cpf = qrot(gi_q, cp - gp);
      cqf = qmul(gi_q, cq);
      gpf = make_f3(0.f, 0.f, 0.f);
      gqf = Quat{0.f, 0.f, 0.f, 1.f};
    }
    const f3 dl = cpf - gpf;
    const f3 wd = make_f3(dl.x * wpos.x, dl.y * wpos.y, dl.z * wpos.z);
    float pd = 0.5f * pw * dot(wd, wd);
    f3 pg = make_f3(pw * wpos.x * wpos.x * dl.x, pw * wpos.y * wpos.y * dl.y, pw * wpos.z * wpos.z * dl.z);
    if (pd < tol_p) { pd = 0.0f; pg = make_f3(0.f, 0.f, 0.f); }
    float rd, angle;
    f3 rg;
    rotation_error(cqf, gqf, wrot, rw, tol_r, a.rotation_method, rd, rg, angle);
    const float tot = pd + rd;
    if (best < 0.0f || tot < best) {
      best = tot; best_g = g; best_pd = pd; best_rd = rd; best_angle = angle;
      best_pg = pg; best_rg = rg; best_gq = gq;
    }
  }
  if (project) {
    best_pg = qrot(best_gq, best_pg);
    best_rg = qrot(best_gq, best_rg);
  }
  const Quat qr = qmul(cq, Quat{best_rg.x, best_rg.y, best_rg.z, 0.0f});  // q (x) (omega, 0), :107-126
  ToolPoseResult r;
  r.position_cost = best_pd;
  r.rotation_cost = best_rd;
  r.position_distance = pw > 0.0f ? sqrtf(2.0f * best_pd / pw) : 0.0f;
  r.rotation_distance = best_angle;
  r.position_gradient = best_pg;
  r.quat_rate_wxyz = make_float4(qr.w, qr.x, qr.y, qr.z);
  r.goalset_idx = best_g;
  return r;
}

struct CspacePosArgs {
  float *out_cost, *out_grad_p, *out_grad_tau;
  const float *pos, *effort, *cspace_target;
  const int32_t *cspace_target_idx;
  const float *p_b, *effort_b, *weight, *activation_distance, *cspace_target_weight, *cspace_target_dof_weight;
  const float *squared_l2_reg_weight, *current_position, *current_velocity;
  const int32_t *idxs_current_state;
  const float *v_b, *state_dt;
  int write_grad, batch, horizon, dof;
};

// joint-limit term of the c-space cost for one dof (wp_cspace_position.py:268-300): limits shrunk
// by eta * range, quadratic outside; returns the cost, g = d cost / d position
__device__ __forceinline__ float cspace_bound_term(float cp, float pl, float pu, float w, float &g) {
  g = 0.0f;
  if (cp < pl || cp > pu) {
    const float delta = cp < pl ? cp - pl : cp - pu;
    const float wv = w * delta;
    g = wv;
    return 0.5f * wv * delta;
  }
  return 0.0f;
}

// one (batch, horizon, dof) entry, wp_cspace_position.py:232-362
__device__ __forceinline__ float cspace_position_point(const CspacePosArgs &a, int b, int d, float cp, float ctau,
                                                       float &gp, float &gt) {
  const int dof = a.dof;
  const float eta_p = a.activation_distance[0], eta_tau = a.activation_distance[1];
  const float w = a.weight[0], tau_w = a.weight[1];
  float tl = a.effort_b[d], tu = a.effort_b[dof + d];
  { const float r = tu - tl; tl = tl + eta_tau * r; tu = tu - eta_tau * r; }
  float pl = a.p_b[d], pu = a.p_b[dof + d];
  { const float r = pu - pl; pl = pl + eta_p * r; pu = pu - eta_p * r; }
  const int cur = a.idxs_current_state[b];
  const float dt = a.state_dt[cur];
  float cur_p = 0.0f;
  if (dt > 0.0f) {
    cur_p = a.current_position[(size_t)cur * dof + d];
    pl = fmaxf(pl, cur_p + a.v_b[d] * dt);
    pu = fminf(pu, cur_p + a.v_b[dof + d] * dt);
  }
  float c = 0.0f;
  gt = 0.0f;
  c += cspace_bound_term(cp, pl, pu, w, gp);
  if (tau_w > 0.0f && (ctau < tl || ctau > tu)) {
    const float delta = ctau < tl ? ctau - tl : ctau - tu;
    const float wv = tau_w * delta;
    c += 0.5f * wv * delta;
    gt += wv;
  }
  const float tw = a.cspace_target_weight[0] * a.cspace_target_dof_weight[d];
  if (tw > 0.0f) {
    const float e = cp - a.cspace_target[(size_t)a.cspace_target_idx[b] * dof + d];
    c += tw * e * e;
    gp += 2.0f * tw * e;
  }
  const float vw = a.squared_l2_reg_weight[0] * dt, aw = a.squared_l2_reg_weight[1] * dt * dt;
  if (dt > 0.0f && (vw > 0.0f || aw > 0.0f)) {
    const float vi = (cp - cur_p) / dt;
    if (vw > 0.0f) { c += 0.5f * vw * vi * vi; gp += vw * vi / dt; }
    if (aw > 0.0f) {
      const float ai = (vi - a.current_velocity[(size_t)cur * dof + d]) / dt;
      c += 0.5f * aw * ai * ai;
      gp += aw * ai / (dt * dt);
    }
  }
  return c;
}

// ------------------------------------------------------------------------------------------
// c-space STATE cost (curobo/_src/cost/wp_cspace_state.py:20-287, cost/warp_bound_util.py)
struct CspaceStateArgs {
  float *out_cost, *out_gp, *out_gv, *out_ga, *out_gj, *out_gtau;
  const float *pos, *vel, *acc, *jerk, *effort, *state_dt, *target;
  const int32_t *idxs_target;
  const float *p_b, *v_b, *a_b, *j_b, *effort_b, *weight, *activation_distance, *sql2_weights;
  const float *target_weight, *non_terminal_factor, *target_dof_weight;
  int write_grad, batch, horizon, dof, retime_weights, retime_reg_weights;
};

__device__ __forceinline__ void squared_l2_term(float x, float w, float &c, float &g) {
  const float wv = w * x;
  c += 0.5f * wv * x;
  g += wv;
}
__device__ __forceinline__ void bound_term(float x, const float *lim, int dof, int d, float eta, float w, float &c, float &g) {
  float lo = lim[d], hi = lim[dof + d];
  const float r = hi - lo;
  lo = lo + eta * r;
  hi = hi - eta * r;
  if (x < lo) squared_l2_term(x - lo, w, c, g);
  else if (x > hi) squared_l2_term(x - hi, w, c, g);
}

// one (batch, horizon, dof) entry: x = (position, velocity, acceleration, jerk, effort) -> cost, g[5]
__device__ __forceinline__ float cspace_state_point(const CspaceStateArgs &a, int b, int h, int d, const float (&x)[5],
                                                    float (&g)[5]) {
  const float dt = a.state_dt[b];
  float wb[5], wr[5];
#pragma unroll
  for (int i = 0; i < 5; i++) { wb[i] = a.weight[i]; wr[i] = a.sql2_weights[i]; g[i] = 0.0f; }
  // (the reference writes wp.pow(dt, 2.0) / wp.pow(dt, 3.0); plain products are within an ulp and keep
  // the generic pow expansion out of the fused kernel's register budget)
  const float dt2 = dt * dt, dt3 = dt * dt * dt;
  if (a.retime_weights) { wb[1] = dt * wb[1]; wb[2] = dt2 * wb[2]; wb[3] = dt3 * wb[3]; }
  if (a.retime_reg_weights) { wr[0] = dt * wr[0]; wr[1] = dt2 * wr[1]; wr[2] = dt3 * wr[2]; wr[4] = dt * wr[4]; }
  float c = 0.0f;
  bound_term(x[0], a.p_b, a.dof, d, a.activation_distance[0], wb[0], c, g[0]);
  bound_term(x[1], a.v_b, a.dof, d, a.activation_distance[1], wb[1], c, g[1]);
  bound_term(x[2], a.a_b, a.dof, d, a.activation_distance[2], wb[2], c, g[2]);
  bound_term(x[3], a.j_b, a.dof, d, a.activation_distance[3], wb[3], c, g[3]);
  bound_term(x[4], a.effort_b, a.dof, d, a.activation_distance[4], wb[4], c, g[4]);
  float tw = a.target_weight[0];
  if (h < a.horizon - 1) tw *= a.non_terminal_factor[0];
  if (tw > 0.0f) {
    tw *= a.target_dof_weight[d];
    const float e = x[0] - a.target[(size_t)a.idxs_target[b] * a.dof + d];
    c += tw * e * e;
    g[0] += 2.0f * tw * e;
  }
  squared_l2_term(x[1], wr[0], c, g[1]);
  squared_l2_term(x[2], wr[1], c, g[2]);
  squared_l2_term(x[3], wr[2], c, g[3]);
  squared_l2_term(x[4], wr[3], c, g[4]);
  if (wr[4] > 0.0f) {  // aggregate_energy_regularization
    const float e = x[4] * x[1] * dt;
    c += wr[4] * e * e;
    g[4] += 2.0f * wr[4] * e * x[1] * dt;
    g[1] += 2.0f * wr[4] * e * x[4] * dt;
  }
  return c;
}

}  // namespace curobo_hip
