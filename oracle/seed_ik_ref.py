"""CPU restatement of the reference's Levenberg-Marquardt seed-IK iteration (TEST INFRASTRUCTURE:
only tests/ may import this; the product path is curobo_amd/solver/seed_ik.py on the HIP kernels).

Follows, with the oracle's FK / Jacobian / FK-VJP / tool-pose / LM-step restatements as building blocks:
  curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:128-231  error + Jacobian of a configuration
      :233-290  pose block: FK with Jacobian, ToolPoseCost (weights [pw, ow], unit axes weights, zero
                tolerance, use_lie_group=False), J^T e through the FK backward (use_backward=True)
      :292-305  position / orientation error = max over the tool frames, error norm = sum of the cost
      :338-387  joint-limit block (diagonal Jacobian rows)
      :464-495  combination
  curobo/_src/solver/seed_ik/seed_iteration_state_manager.py:74-260  state update
  curobo/_src/solver/seed_ik/seed_ik_solver.py:291-330,384-437     iteration / solve loop
The state update is pinned bit-exactly by the reference's own SeedIterationStateManager run on CPU
(tests/golden/seed_ik_update_golden.npz, tests/golden/make_seed_ik_golden.py), the joint-limit block
(with and without velocity clamping of the bounds) by the reference's own
SeedIKErrorCalculator._compute_joint_limit_errors (tests/golden/seed_ik_limits_golden.npz).  Parity of the other
pieces is pinned where they are defined (oracle/curobo_oracle.c); the LM step is pinned by the reference's own Warp
tile kernel run on the CPU through the stand-in of tests/golden/warp_emulator (tests/golden/lm_warp_golden.npz,
bit-equal) and against numpy.linalg.solve in tests/test_oracle_linalg.py.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np


@dataclass
class SeedIKRefCfg:  # names and defaults: solver/seed_ik/seed_ik_solver_cfg.py:25-94
    max_iterations: int = 16
    inner_iterations: int = 4
    position_tolerance: float = 0.005
    orientation_tolerance: float = 0.05
    convergence_position_tolerance: float = 0.00001
    convergence_orientation_tolerance: float = 0.00001
    convergence_joint_limit_weight: float = 1.0
    lambda_initial: float = 0.2
    lambda_factor: float = 2.0
    lambda_max: float = 1.0e10
    lambda_min: float = 1e-5
    joint_limit_margin: float = 0.001
    joint_limit_weight: float = 1.0
    rho_min: float = 1e-3
    position_weight: float = 1.0
    orientation_weight: float = 1.0


def action_bounds(model: Dict[str, np.ndarray], cfg: SeedIKRefCfg):
    lo, hi = np.asarray(model["joint_limits_position"], np.float32)
    margin = (hi - lo) * np.float32(cfg.joint_limit_margin)
    return (lo + margin).astype(np.float32), (hi - margin).astype(np.float32)


def joint_limit_block(q, lo, hi, weight, current_position=None, dt=None, velocity_limits=None):
    """seed_ik_error_calculator.py:338-387: (J^T e contribution [n, D], Jacobian diagonal [n, D], summed
    error [n]).  With ``current_position`` [n, D], ``dt`` [n] and ``velocity_limits`` [2, D] (lower row
    negative) the bounds are tightened to what one step of ``dt`` can reach (:355-363)."""
    q = np.asarray(q, np.float32)
    lo = np.broadcast_to(np.asarray(lo, np.float32), q.shape)
    hi = np.broadcast_to(np.asarray(hi, np.float32), q.shape)
    if current_position is not None and dt is not None:
        v = np.asarray(velocity_limits, np.float32)
        dtc = np.asarray(dt, np.float32).reshape(-1, 1)
        cp = np.asarray(current_position, np.float32)
        lo = np.maximum(lo, cp + v[0] * dtc)
        hi = np.minimum(hi, cp + v[1] * dtc)
    uv, lv = np.maximum(q - hi, np.float32(0)), np.maximum(lo - q, np.float32(0))
    w = np.float32(weight)
    err = w * (lv + uv)
    diag = w * (np.where(lv > 0, -1.0, 0.0) + np.where(uv > 0, 1.0, 0.0)).astype(np.float32)
    return (diag * err).astype(np.float32), diag, err.sum(-1).astype(np.float32)


def evaluate(orc, model, cfg: SeedIKRefCfg, q, goal_position, goal_quat, idxs_goal, current_position=None, dt=None):
    """error + Jacobian of configurations q[n, D] against goals [P, T, G, 3|4] (seed_ik_error_calculator.py:128-231);
    ``current_position`` / ``dt``: velocity clamping of the joint-limit bounds with the model's velocity limits"""
    q = np.ascontiguousarray(q, np.float32)
    n, D = q.shape
    T = model["tool_frame_map"].shape[0]
    fk = orc.kinematics_forward(q, model, compute_jacobian=True, compute_spheres=False)
    one6 = np.ones(6, np.float32)
    tp = orc.tool_pose_distance(
        fk["link_pos"].reshape(n, 1, T, 3), fk["link_quat"].reshape(n, 1, T, 4), goal_position, goal_quat, idxs_goal,
        np.array([cfg.position_weight, cfg.orientation_weight], np.float32), np.tile(one6, T), np.tile(one6, T),
        np.zeros(2 * T, np.float32), np.zeros(2 * T, np.float32), np.zeros(T, np.uint8), 0)
    pose_jte = orc.kinematics_backward(model, fk["cumul_mat"], None, tp["position_gradient"].reshape(n, T, 3),
                                       tp["rotation_gradient"].reshape(n, T, 4))
    lo, hi = action_bounds(model, cfg)
    jl_jte, diag, jl_sum = joint_limit_block(q, lo, hi, cfg.joint_limit_weight, current_position, dt,
                                             model.get("joint_limits_velocity"))
    J = np.zeros((n, 6 * T + D, D), np.float32)
    J[:, : 6 * T] = fk["jacobian"].reshape(n, 6 * T, D)
    J[:, 6 * T + np.arange(D), np.arange(D)] = diag
    return {
        "joint_position": q,
        "jacobian": J,
        "jTerror": (pose_jte + jl_jte).astype(np.float32),
        "error_norm": (tp["distance"].reshape(n, -1).sum(-1) + jl_sum).astype(np.float32),
        "position_errors": tp["position_distance"].reshape(n, T).max(-1),
        "orientation_errors": tp["rotation_distance"].reshape(n, T).max(-1),
        "pose_jacobian": fk["jacobian"].reshape(n, 6 * T, D), "pose_jTerror": pose_jte,
        "pose_cost": tp["distance"].reshape(n, T, 2) if tp["distance"].shape[-1] == 2 * T else tp["distance"],
        "position_distance": tp["position_distance"].reshape(n, T), "rotation_distance": tp["rotation_distance"].reshape(n, T),
    }


def update_state(cur, cand, pred_reduction, lo, hi, cfg: SeedIKRefCfg):
    """seed_iteration_state_manager.py:74-260 (cur carries lambda_damping)"""
    rho = (cur["error_norm"] - cand["error_norm"]) / (pred_reduction + np.float32(1e-8))
    acc = rho >= cfg.rho_min
    lam = np.where(acc, cur["lambda_damping"] / cfg.lambda_factor, cur["lambda_damping"] * cfg.lambda_factor)
    lam = np.clip(lam, cfg.lambda_min, cfg.lambda_max).astype(np.float32)
    sel = {k: np.where(acc.reshape((-1,) + (1,) * (cand[k].ndim - 1)), cand[k], cur[k])
           for k in ("joint_position", "jTerror", "jacobian", "position_errors", "orientation_errors")}
    ok = (sel["position_errors"] < cfg.convergence_position_tolerance) & (
        sel["orientation_errors"] < cfg.convergence_orientation_tolerance)
    if cfg.convergence_joint_limit_weight > 0:
        ok &= np.all((sel["joint_position"] > lo) & (sel["joint_position"] < hi), axis=-1)
    return {**sel, "lambda_damping": lam, "error_norm": cand["error_norm"], "success": ok, "improvement": acc}


def solve(orc, model, cfg: SeedIKRefCfg, seeds, goal_position, goal_quat, idxs_goal):
    """all iterations, no early exit (seed_ik_solver.py:384-437 with batch_success_threshold never met)"""
    lo, hi = action_bounds(model, cfg)
    st = evaluate(orc, model, cfg, seeds, goal_position, goal_quat, idxs_goal)
    st["lambda_damping"] = np.full(seeds.shape[0], cfg.lambda_initial, np.float32)
    for _ in range(cfg.max_iterations):
        q_new, pred = orc.lm_step(st["jacobian"], st["jTerror"], st["lambda_damping"], st["joint_position"])
        cand = evaluate(orc, model, cfg, q_new, goal_position, goal_quat, idxs_goal)
        st = update_state(st, cand, pred, lo, hi, cfg)
    ok = (st["position_errors"] < cfg.position_tolerance) & (st["orientation_errors"] < cfg.orientation_tolerance)
    lim_lo, lim_hi = np.asarray(model["joint_limits_position"], np.float32)
    ok &= np.all((st["joint_position"] > lim_lo) & (st["joint_position"] < lim_hi), axis=-1)
    st["final_success"] = ok
    return st
