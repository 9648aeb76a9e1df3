"""Oracle composition of the FULL trajopt rollout (TEST INFRASTRUCTURE ONLY): knots -> B-spline -> FK -> tool-pose goal
cost + c-space STATE cost (+ joint-torque limits through RNEA and its VJP) + self collision (+ scene collision) -> cost per
trajectory and d cost / d knots, chained from the C oracle's stages the way ``TrajOptRollout.evaluate_action`` chains the
HIP launches (reference call stack: rollout/rollout_robot.py:252-263, 537-587)."""

import numpy as np


def trajopt_cost_and_gradient(oracle, model, cfg, knots, start, *, goal_position=None, goal_quat=None, idxs_goal=None,
                              scene_arrays=None, effort_limit=None, sweep=None, scene_spheres=None):
    """``cfg``: a ``TrajOptRolloutCfg``.  Goals default to what a fresh ``TrajOptRollout`` holds (origin, identity
    quaternion, one goal).  ``scene_spheres`` [B, H, S, 4]: evaluate the scene stage on these (the device's own FK output) instead
    of the oracle's -- the swept cost is discontinuous in the sphere positions (a sphere that does not move counts its centre
    up to three times, wp_sweep_collision_kernel.py:197-203), so the last bit of an FK decides.  Returns dict(cost [B], grad_knots [B, nk, D], tau [B*H, D] or None, parts)."""
    md = model.as_dict()
    B, nk, D = knots.shape
    H = cfg.padded_horizon
    T = int(md["tool_frame_map"].shape[0])
    zeros = np.zeros((1, D), np.float32)
    st = {"position": np.asarray(start, np.float32).reshape(1, D), "velocity": zeros, "acceleration": zeros, "jerk": zeros}
    gl = {k: zeros for k in st}
    i0 = np.zeros(B, np.int32)
    dt = np.array([cfg.traj_dt], np.float32)
    imp = np.zeros(1, np.uint8)
    s = oracle.bspline_forward(knots, st, gl, i0, i0, dt, imp, H, cfg.bspline_degree)
    fk = oracle.kinematics_forward(s["position"].reshape(B * H, D), md, horizon=H)
    if goal_position is None:
        goal_position = np.zeros((1, T, 1, 3), np.float32)
        goal_quat = np.zeros((1, T, 1, 4), np.float32)
        goal_quat[..., 0] = 1.0
        idxs_goal = np.zeros(B, np.int32)
    tol = np.tile(np.asarray(cfg.pose_convergence_tolerance, np.float32), (T, 1))
    pose = oracle.tool_pose_distance(fk["link_pos"].reshape(B, H, T, 3), fk["link_quat"].reshape(B, H, T, 4), goal_position, goal_quat,
                                     idxs_goal, np.array(cfg.pose_weight, np.float32), np.ones((T, 6), np.float32),
                                     np.full((T, 6), cfg.non_terminal_pose_factor, np.float32), tol, tol, np.zeros(T, np.uint8),
                                     cfg.rotation_method)
    ones = np.ones(D, np.float32)
    lim = {"position": model.joint_limits_position.astype(np.float32), "velocity": model.joint_limits_velocity.astype(np.float32),
           "acceleration": np.stack([-cfg.max_acceleration * ones, cfg.max_acceleration * ones]),
           "jerk": np.stack([-cfg.max_jerk * ones, cfg.max_jerk * ones])}
    extra, tau, cache = {}, None, None
    flat = lambda a: np.ascontiguousarray(a.reshape(B * H, D))  # noqa: E731
    grav = np.array(cfg.gravity, np.float32)
    if cfg.use_torque_limits:
        tau, cache = oracle.rnea_forward(flat(s["position"]), flat(s["velocity"]), flat(s["acceleration"]), md, gravity=grav)
        cap = np.abs(np.asarray(effort_limit if effort_limit is not None else cfg.effort_limit, np.float32))
        lim["effort"] = np.stack([-cap, cap])
        extra["effort"] = tau.reshape(B, H, D)
    cs = oracle.cspace_state_cost(s["position"], s["velocity"], s["acceleration"], s["jerk"], np.full(B, cfg.traj_dt, np.float32),
                                  lim, cfg.cspace_weight, cfg.cspace_activation_distance, cfg.cspace_regularization,
                                  retime_weights=cfg.retime_weights, retime_regularization_weights=cfg.retime_regularization_weights,
                                  **extra)
    if cfg.use_torque_limits:
        gr = oracle.rnea_backward(flat(cs["grad_effort"]), flat(s["position"]), flat(s["velocity"]), cache, md, gravity=grav)
        for key, g in zip(("grad_position", "grad_velocity", "grad_acceleration"), gr):
            cs[key] = cs[key] + g.reshape(B, H, D)
    sph = fk["robot_spheres"].reshape(B, H, -1, 4)
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, cfg.self_collision_weight)
    cost = pose["distance"].sum((1, 2)).astype(np.float64) + cs["cost"].sum((1, 2)) + sc["distance"].reshape(B, H).sum(1)
    gs = sc["gradient"].reshape(B * H, -1, 4).copy()
    wc = None
    if scene_arrays is not None:
        use_sweep = cfg.use_sweep if sweep is None else sweep
        wc = oracle.scene_collision(sph if scene_spheres is None else np.asarray(scene_spheres, np.float32).reshape(sph.shape), scene_arrays, cfg.scene_collision_weight, cfg.scene_activation_distance, sweep=use_sweep,
                                    enable_speed_metric=use_sweep and cfg.use_speed_metric, speed_dt=cfg.traj_dt)
        cost = cost + wc["distance"].sum((1, 2))
        gs = gs + wc["gradient"].reshape(B * H, -1, 4) * np.array([1, 1, 1, 0], np.float32)
    gq = oracle.kinematics_backward(md, fk["cumul_mat"], gs, pose["position_gradient"].reshape(B * H, T, 3),
                                    pose["rotation_gradient"].reshape(B * H, T, 4), horizon=H).reshape(B, H, D)
    gk = oracle.bspline_backward(gq + cs["grad_position"], cs["grad_velocity"], cs["grad_acceleration"], cs["grad_jerk"], dt, i0,
                                 imp, nk, cfg.bspline_degree)
    return {"cost": cost, "grad_knots": gk, "tau": tau, "pose": pose, "cspace": cs, "self": sc, "scene": wc, "state": s, "fk": fk}
