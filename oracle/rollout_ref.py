"""Oracle composition of the rollout hot path (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

knots -> B-spline -> FK -> self + scene collision -> per-trajectory cost and the VJP back to the
knots, by chaining the C oracle's kernels exactly as the reference chains its autograd
functions.
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .oracle import Oracle


def rollout_cost_and_gradient(
    orc: Oracle,
    model: Dict[str, np.ndarray],
    scene: Optional[Dict[str, np.ndarray]],
    knots: np.ndarray,
    start_position: np.ndarray,
    *,
    interpolation_steps: int = 2,
    degree: int = 3,
    traj_dt: float = 0.05,
    self_collision_weight: float = 10000.0,
    scene_collision_weight: float = 100000.0,
    activation_distance: float = 0.0025,
    use_sweep: bool = True,
    use_speed_metric: bool = True,
    env_query_idx: Optional[np.ndarray] = None,
) -> Dict[str, np.ndarray]:
    b, nk, d = knots.shape
    ph = (nk + degree + 1) * interpolation_steps + 1
    zeros = np.zeros((1, d), np.float32)
    start = {"position": start_position.reshape(1, d).astype(np.float32), "velocity": zeros,
             "acceleration": zeros, "jerk": zeros}
    goal = {k: zeros for k in start}
    idx0 = np.zeros((b,), np.int32)
    dt = np.array([traj_dt], np.float32)
    imp = np.zeros((1,), np.uint8)
    state = orc.bspline_forward(knots, start, goal, idx0, idx0, dt, imp, ph, degree)
    q = state["position"].reshape(b * ph, d)
    fk = orc.kinematics_forward(q, model, horizon=ph)
    S = fk["robot_spheres"].shape[1]
    sph = fk["robot_spheres"].reshape(b, ph, S, 4)
    sc = orc.self_collision(sph, model["sphere_padding"], model["collision_pairs"], self_collision_weight)
    grad_sph = sc["gradient"].reshape(b, ph, S, 4).copy()
    self_cost = sc["distance"].reshape(b, ph)
    scene_cost = np.zeros((b, ph, S), np.float32)
    if scene is not None:
        wc = orc.scene_collision(sph, scene, scene_collision_weight, activation_distance, sweep=use_sweep,
                                 enable_speed_metric=use_sweep and use_speed_metric, speed_dt=traj_dt,
                                 env_query_idx=env_query_idx, use_multi_env=env_query_idx is not None)
        scene_cost = wc["distance"]
        grad_sph[..., :3] += wc["gradient"][..., :3]
    cost = orc.trajectory_cost_sum(self_cost, scene_cost)
    grad_q = orc.kinematics_backward(model, fk["cumul_mat"], grad_sph.reshape(b * ph, S, 4), horizon=ph)
    gz = np.zeros((b, ph, d), np.float32)
    grad_knots = orc.bspline_backward(grad_q.reshape(b, ph, d), gz, gz, gz, dt, idx0, imp, nk, degree)
    return {"cost": cost, "grad_knots": grad_knots, "position": state["position"], "robot_spheres": sph,
            "self_cost": self_cost, "scene_cost": scene_cost, "grad_q": grad_q.reshape(b, ph, d),
            "link_pos": fk["link_pos"], "link_quat": fk["link_quat"]}
