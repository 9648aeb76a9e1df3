"""Pins for the tool-pose and c-space cost oracles.  The key pin: the (tool-pose cost, FK VJP)
PAIR must reproduce the finite-difference derivative of the pose cost w.r.t. the joints -- this is
what fixes the orientation-gradient convention of both (reference cost/wp_tool_pose.py:107-126 +
common/quaternion_util.cuh:86-102)."""

import numpy as np
import pytest

from conftest import sample_q


def _pose_cost(oracle, model, q, goal_pos, goal_quat, w=(10000.0, 500.0), method=0, project=0, tol=1e-8):
    md = model.as_dict()
    n = q.shape[0]
    fk = oracle.kinematics_forward(q, md)
    T = fk["link_pos"].shape[1]
    out = oracle.tool_pose_distance(
        fk["link_pos"].reshape(n, 1, T, 3), fk["link_quat"].reshape(n, 1, T, 4), goal_pos, goal_quat,
        np.zeros(n, np.int32), np.array(w, np.float32), np.ones((T, 6), np.float32), np.ones((T, 6), np.float32),
        np.full((T, 2), tol, np.float32), np.full((T, 2), tol, np.float32), np.full((T,), project, np.uint8), method)
    return fk, out


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("project", [0, 1])
@pytest.mark.parametrize("part", ["position", "rotation"])
def test_pose_cost_gradient_through_fk_matches_finite_differences(part, project, method, oracle, franka):
    """d(pose cost)/dq via (tool-pose gradient -> FK VJP) vs central differences.

    Position part: exact for every method.  Rotation part: the Lie-group methods (1, 2) are exact
    (cost = w (theta/2)^2); the axis-angle method (0, the IK default) emits 2 w theta * axis for the
    cost w theta^2 and the FK VJP halves the quaternion rate, so the reference's gradient is
    exactly HALF of the true derivative (`# quat_rate = 0.5 * quat_rate` is commented out in
    wp_tool_pose.py:125) -- restated as is and pinned as such."""
    md = franka.as_dict()
    q = sample_q(franka, 6, seed=3, scale=0.6)
    fg = oracle.kinematics_forward(sample_q(franka, 1, seed=4, scale=0.6), md)
    goal_pos = fg["link_pos"].reshape(1, 1, 1, 3)
    goal_quat = fg["link_quat"].reshape(1, 1, 1, 4)
    w = (100.0, 0.0) if part == "position" else (0.0, 30.0)
    fk, out = _pose_cost(oracle, franka, q, goal_pos, goal_quat, w, method, project)
    assert (out["distance"].sum(-1)[:, 0] > 0).all()
    got = oracle.kinematics_backward(md, fk["cumul_mat"], None, out["position_gradient"].reshape(6, 1, 3),
                                     out["rotation_gradient"].reshape(6, 1, 4))
    eps = 1e-3
    fd = np.zeros((6, 7))
    for j in range(7):
        dq = np.zeros_like(q)
        dq[:, j] = eps
        c1 = _pose_cost(oracle, franka, q + dq, goal_pos, goal_quat, w, method, project)[1]["distance"].sum(-1)[:, 0]
        c0 = _pose_cost(oracle, franka, q - dq, goal_pos, goal_quat, w, method, project)[1]["distance"].sum(-1)[:, 0]
        fd[:, j] = (c1.astype(np.float64) - c0) / (2 * eps)
    if part == "rotation" and method == 0:
        fd = 0.5 * fd
    np.testing.assert_allclose(got, fd, rtol=3e-2, atol=3e-2 * np.abs(fd).max())


def test_pose_cost_zero_at_goal_and_goalset_argmin(oracle, franka):
    md = franka.as_dict()
    q = sample_q(franka, 3, seed=1)
    fk = oracle.kinematics_forward(q, md)
    far = oracle.kinematics_forward(sample_q(franka, 3, seed=2), md)
    # goal set of 2 per problem: [far pose, own pose]  -> index 1 wins with zero cost
    gp = np.stack([far["link_pos"][:, 0], fk["link_pos"][:, 0]], 1).reshape(3, 1, 2, 3)
    gq = np.stack([far["link_quat"][:, 0], fk["link_quat"][:, 0]], 1).reshape(3, 1, 2, 4)
    out = oracle.tool_pose_distance(
        fk["link_pos"].reshape(3, 1, 1, 3), fk["link_quat"].reshape(3, 1, 1, 4), gp, gq, np.arange(3, dtype=np.int32),
        np.array([10000.0, 500.0], np.float32), np.ones((1, 6), np.float32), np.ones((1, 6), np.float32),
        np.full((1, 2), 1e-8, np.float32), np.full((1, 2), 1e-8, np.float32), np.zeros(1, np.uint8), 0)
    assert (out["goalset_idx"] == 1).all()
    np.testing.assert_allclose(out["distance"], 0.0, atol=1e-9)
    np.testing.assert_array_equal(out["position_gradient"], 0.0)
    np.testing.assert_allclose(out["rotation_gradient"], 0.0, atol=1e-4)


def test_pose_cost_values(oracle):
    """position: 0.5 w |W d|^2 ; rotation (axis-angle): w * theta^2 for a pure rotation about z"""
    cp = np.array([[[[0.1, 0.2, 0.3]]]], np.float32)
    th = 0.4
    cq = np.array([[[[np.cos(th / 2), 0, 0, np.sin(th / 2)]]]], np.float32)
    gp = np.zeros((1, 1, 1, 3), np.float32)
    gq = np.array([[[[1, 0, 0, 0]]]], np.float32)
    out = oracle.tool_pose_distance(cp, cq, gp, gq, np.zeros(1, np.int32), np.array([2.0, 3.0], np.float32),
                                    np.ones((1, 6), np.float32), np.ones((1, 6), np.float32), np.zeros((1, 2), np.float32),
                                    np.zeros((1, 2), np.float32), np.zeros(1, np.uint8), 0)
    np.testing.assert_allclose(out["distance"][0, 0], [0.5 * 2.0 * 0.14, 3.0 * th * th], rtol=1e-5)
    np.testing.assert_allclose(out["position_distance"][0, 0, 0], np.sqrt(0.14), rtol=1e-5)
    np.testing.assert_allclose(out["rotation_distance"][0, 0, 0], th, rtol=1e-5)
    np.testing.assert_allclose(out["position_gradient"][0, 0, 0], 2.0 * np.array([0.1, 0.2, 0.3]), rtol=1e-5)


def test_cspace_position_bounds(oracle, franka):
    lo, hi = franka.joint_limits_position.astype(np.float32)
    p_b = np.stack([lo, hi])
    rng = hi - lo
    pos = np.stack([0.5 * (lo + hi), lo - 0.1, hi + 0.2, lo + 0.005 * rng]).reshape(4, 1, 7).astype(np.float32)
    out = oracle.cspace_position_cost(pos, p_b, np.array([5000.0, 0.0], np.float32), np.array([0.01, 0.01], np.float32))
    c, g = out["cost"][:, 0], out["grad_position"][:, 0]
    assert (c[0] == 0).all() and (g[0] == 0).all()
    # activation distance shrinks the limits by 1% of the range on each side
    np.testing.assert_allclose(c[1], 0.5 * 5000.0 * (0.1 + 0.01 * rng) ** 2, rtol=1e-4)
    np.testing.assert_allclose(g[1], -5000.0 * (0.1 + 0.01 * rng), rtol=1e-4)
    np.testing.assert_allclose(g[2], 5000.0 * (0.2 + 0.01 * rng), rtol=1e-4)
    np.testing.assert_allclose(c[3], 0.5 * 5000.0 * (0.005 * rng) ** 2, rtol=1e-3)


def test_cspace_position_regularisers(oracle):
    d = 3
    p_b = np.array([[-10.0] * d, [10.0] * d], np.float32)
    pos = np.array([[[0.3, -0.2, 0.1]]], np.float32)
    cur = np.array([[0.1, 0.1, 0.1]], np.float32)
    curv = np.array([[1.0, 0.0, -1.0]], np.float32)
    dt = 0.1
    out = oracle.cspace_position_cost(
        pos, p_b, np.array([1.0, 0.0], np.float32), np.array([0.0, 0.0], np.float32),
        squared_l2_reg_weight=(2.0, 3.0), current_position=cur, current_velocity=curv, idxs_current_state=np.zeros(1),
        v_b=np.array([[-100.0] * d, [100.0] * d], np.float32), state_dt=np.array([dt], np.float32),
        cspace_target=np.array([[0.0, 0.0, 0.0]], np.float32), cspace_target_idx=np.zeros(1), cspace_target_weight=4.0)
    v = (pos[0, 0] - cur[0]) / dt
    a = (v - curv[0]) / dt
    want = 4.0 * pos[0, 0] ** 2 + 0.5 * (2.0 * dt) * v * v + 0.5 * (3.0 * dt * dt) * a * a
    np.testing.assert_allclose(out["cost"][0, 0], want, rtol=1e-5)
    wantg = 8.0 * pos[0, 0] + (2.0 * dt) * v / dt + (3.0 * dt * dt) * a / (dt * dt)
    np.testing.assert_allclose(out["grad_position"][0, 0], wantg, rtol=1e-5)


def _state_case(rng, b=5, h=6, d=4):
    x = {k: rng.normal(size=(b, h, d)).astype(np.float32) * s for k, s in
         (("pos", 2.0), ("vel", 3.0), ("acc", 8.0), ("jerk", 30.0), ("effort", 40.0))}
    lim = {"position": np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * 1.5,
           "velocity": np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * 2.0,
           "acceleration": np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * 6.0,
           "jerk": np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * 25.0,
           "effort": np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * 30.0}
    return x, lim


def test_cspace_state_cost_gradients_are_exact_derivatives(oracle):
    """every term is a piecewise quadratic of its own input (wp_cspace_state.py:170-260): the
    returned gradients must be the central finite differences of the summed cost"""
    rng = np.random.default_rng(0)
    x, lim = _state_case(rng)
    b, h, d = x["pos"].shape
    kw = dict(state_dt=np.full(b, 0.05, np.float32), limits=lim, weight=[50.0, 20.0, 5.0, 1.0, 2.0],
              activation_distance=[0.05, 0.1, 0.1, 0.1, 0.1], sql2_weights=[0.3, 0.2, 0.1, 0.05, 0.4],
              target=rng.normal(size=(2, d)).astype(np.float32), idxs_target=rng.integers(0, 2, size=b),
              target_weight=3.0, non_terminal_factor=0.25, target_dof_weight=rng.uniform(0.5, 1, size=d).astype(np.float32))
    ref = oracle.cspace_state_cost(x["pos"], x["vel"], x["acc"], x["jerk"], effort=x["effort"], **kw)
    assert (ref["cost"] > 0).all()
    eps = 1e-2
    names = (("pos", "grad_position"), ("vel", "grad_velocity"), ("acc", "grad_acceleration"), ("jerk", "grad_jerk"),
             ("effort", "grad_effort"))
    for key, gname in names:
        xp, xm = dict(x), dict(x)
        xp[key] = x[key] + eps
        xm[key] = x[key] - eps
        cp = oracle.cspace_state_cost(xp["pos"], xp["vel"], xp["acc"], xp["jerk"], effort=xp["effort"], **kw)["cost"]
        cm = oracle.cspace_state_cost(xm["pos"], xm["vel"], xm["acc"], xm["jerk"], effort=xm["effort"], **kw)["cost"]
        fd = (cp.astype(np.float64) - cm) / (2 * eps)
        # a bound kink inside [x - eps, x + eps] makes the FD a blend: compare away from kinks
        g = ref[gname]
        ok = np.abs(fd - g) <= 2e-2 * np.maximum(1.0, np.abs(g))
        assert ok.mean() > 0.9, (key, ok.mean())


def test_cspace_state_cost_semantics(oracle):
    rng = np.random.default_rng(1)
    x, lim = _state_case(rng)
    b, h, d = x["pos"].shape
    z5 = [0.0] * 5
    base = dict(state_dt=np.full(b, 0.1, np.float32), limits=lim, activation_distance=z5)
    # inside all limits and without regularisation the cost is zero
    small = {k: v * 0.0 for k, v in x.items()}
    r0 = oracle.cspace_state_cost(small["pos"], small["vel"], small["acc"], small["jerk"], weight=[1.0] * 5, sql2_weights=z5, **base)
    assert np.all(r0["cost"] == 0.0)
    # position bound only: 0.5 w (x - limit)^2 outside, limits shrunk by eta * range
    r1 = oracle.cspace_state_cost(x["pos"], small["vel"], small["acc"], small["jerk"], weight=[10.0, 0, 0, 0, 0], sql2_weights=z5,
                                  state_dt=base["state_dt"], limits=lim, activation_distance=[0.1, 0, 0, 0, 0])
    lo, hi = -1.5 + 0.1 * 3.0, 1.5 - 0.1 * 3.0
    delta = np.where(x["pos"] < lo, x["pos"] - lo, np.where(x["pos"] > hi, x["pos"] - hi, 0.0))
    np.testing.assert_allclose(r1["cost"], 0.5 * 10.0 * delta ** 2, rtol=1e-5, atol=1e-6)
    # retimed weights scale with dt, dt^2, dt^3
    r2 = oracle.cspace_state_cost(small["pos"], x["vel"], x["acc"], x["jerk"], weight=z5, sql2_weights=[1.0, 1.0, 1.0, 0, 0],
                                  retime_regularization_weights=True, **base)
    want = 0.5 * (0.1 * x["vel"] ** 2 + 0.1 ** 2 * x["acc"] ** 2 + 0.1 ** 3 * x["jerk"] ** 2)
    np.testing.assert_allclose(r2["cost"], want, rtol=2e-5, atol=1e-5)
    # the joint target is down-weighted before the last step
    tgt = np.zeros((1, d), np.float32)
    r3 = oracle.cspace_state_cost(x["pos"], small["vel"], small["acc"], small["jerk"], weight=z5, sql2_weights=z5, target=tgt,
                                  target_weight=2.0, non_terminal_factor=0.5, **base)
    np.testing.assert_allclose(r3["cost"][:, -1], 2.0 * x["pos"][:, -1] ** 2, rtol=1e-5)
    np.testing.assert_allclose(r3["cost"][:, 0], 1.0 * x["pos"][:, 0] ** 2, rtol=1e-5)


def test_rotation_distance_matches_reference_metrics_golden(oracle):
    """the tool-pose cost's rotation distance: method 0 = rotation angle of the relative rotation
    (reference geom/quaternion.py angular_distance_axis_angle), method 1 = acos|<q1, q2>| (= phi3 * pi/2)
    -- against outputs of the reference's own torch functions (tests/golden/make_rotation_metric_golden.py)"""
    import os

    from conftest import GOLDEN_DIR

    g = np.load(os.path.join(GOLDEN_DIR, "rotation_metric_golden.npz"))
    cur, goal = g["current_quat"], g["goal_quat"]
    n = cur.shape[0]
    p = np.zeros((n, 1, 1, 3), np.float32)
    one6, z2 = np.ones((1, 6), np.float32), np.zeros((1, 2), np.float32)
    for method, want in ((0, g["axis_angle"]), (1, g["phi3"] * np.float32(np.pi / 2))):
        out = oracle.tool_pose_distance(p, cur.reshape(n, 1, 1, 4), p, goal.reshape(n, 1, 1, 4), np.arange(n, dtype=np.int32),
                                        np.array([1.0, 1.0], np.float32), one6, one6, z2, z2, np.zeros(1, np.uint8), method)
        got = out["rotation_distance"].reshape(n)
        # acos / atan2 near |dot| = 1 lose digits: 1e-3 rad absolute there, 1e-6 elsewhere
        near = want < 1e-2
        np.testing.assert_allclose(got[~near], want[~near], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(got[near], want[near], atol=1e-3)
    assert (g["axis_angle"][:16] < 1e-3).all() and (g["axis_angle"][16:24] > 1e-4).all()
