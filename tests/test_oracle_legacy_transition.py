"""Pins for the legacy POSITION / ACCELERATION control-space oracle
(kernels/trajectory/legacy/*.cuh).  The reference ships no numeric fixtures for them, so the
restatement is pinned by what the kernels are by construction: five-point finite-difference
stencils (exact for quartic / cubic / quartic polynomials), the constant-acceleration
back-extrapolation of the start state, goal replication, the backward pass being the transpose of
the forward stencils for interior actions, and semi-implicit Euler integration.  (Since round 2 the reference's own
kernels run on the CPU next to the oracle as well: tests/test_reference_cuda_kernels.py.)"""

import numpy as np
import pytest

B, H, DOF, DT = 3, 24, 4, 0.05


def _poly(c):
    f = lambda t: c[0] + c[1] * t + c[2] * t ** 2 + c[3] * t ** 3  # noqa: E731
    d1 = lambda t: c[1] + 2 * c[2] * t + 3 * c[3] * t ** 2  # noqa: E731
    d2 = lambda t: 2 * c[2] + 6 * c[3] * t  # noqa: E731
    return f, d1, d2


def _forward(oracle, u, start, goal, use_goal):
    b = u.shape[0]
    z = np.zeros(b, np.int32)
    return oracle.differentiation_position_forward(u, start, goal, z, z, np.array([DT], np.float32),
                                                   np.array([use_goal], np.uint8))


def test_forward_reproduces_cubic_and_its_derivatives(oracle):
    """action i sits at time (i+1) dt, the start state at t = 0; point h reports time (h-1) dt"""
    c = (0.3, -1.2, 0.8, 0.5)
    f, d1, d2 = _poly(c)
    u = np.broadcast_to(f((np.arange(H - 4) + 1) * DT)[None, :, None], (B, H - 4, DOF)).astype(np.float32).copy()
    start = {"position": np.full((1, DOF), f(0.0), np.float32), "velocity": np.full((1, DOF), d1(0.0), np.float32),
             "acceleration": np.full((1, DOF), d2(0.0), np.float32)}
    out = _forward(oracle, u, start, np.zeros((1, DOF), np.float32), 0)
    h = np.arange(4, H - 5)
    t = (h - 1) * DT
    np.testing.assert_allclose(out["position"][0, h, 0], f(t), atol=2e-6)
    np.testing.assert_allclose(out["velocity"][0, h, 0], d1(t), atol=2e-4)
    np.testing.assert_allclose(out["acceleration"][0, h, 0], d2(t), atol=2e-2)
    np.testing.assert_allclose(out["jerk"][0, h, 0], 6 * c[3], atol=1.0)
    np.testing.assert_allclose(out["dt"], DT)
    # start boundary: point 1 is the start state itself; the back-extrapolation is exact for a
    # constant-acceleration motion, so a quadratic start is differentiated exactly there too
    np.testing.assert_allclose(out["position"][:, 1], f(0.0), atol=1e-6)


def test_start_extrapolation_formulas(oracle):
    """the three virtual points before the start state are the reference's closed forms
    (differentiation_position_kernel.cuh:93-108 with fixed_jerk = 0): they show up as the
    positions reported at h = 0 (= e(-1)) and inside the first stencils"""
    rng = np.random.default_rng(5)
    u = rng.normal(size=(B, H - 4, DOF)).astype(np.float32)
    start = {k: rng.normal(size=(1, DOF)).astype(np.float32) for k in ("position", "velocity", "acceleration")}
    out = _forward(oracle, u, start, np.zeros((1, DOF), np.float32), 0)
    x0, v0, a0 = (start[k].astype(np.float64) for k in ("position", "velocity", "acceleration"))
    e1 = -1.5 * a0 * DT ** 2 - DT * v0 + x0
    e2 = -2.0 * a0 * DT ** 2 - 2.0 * DT * v0 + x0
    e3 = 1.5 * (-a0 * DT ** 2) - 3.0 * DT * v0 + x0
    np.testing.assert_allclose(out["position"][:, 0], np.broadcast_to(e1, (B, DOF)), atol=1e-6)
    np.testing.assert_allclose(out["position"][:, 1], np.broadcast_to(x0, (B, DOF)), atol=1e-6)
    vel0 = (0.083333333 * e3 - 0.666666667 * e2 + 0.666666667 * x0 - 0.083333333 * u[:, 0]) / DT
    np.testing.assert_allclose(out["velocity"][:, 0], vel0, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("use_goal", [0, 1])
def test_trajectory_ends_at_rest_on_last_action_or_goal(oracle, use_goal):
    rng = np.random.default_rng(1)
    u = rng.normal(size=(B, H - 4, DOF)).astype(np.float32)
    start = {k: rng.normal(size=(1, DOF)).astype(np.float32) * 0.1 for k in ("position", "velocity", "acceleration")}
    goal = rng.normal(size=(1, DOF)).astype(np.float32)
    out = _forward(oracle, u, start, goal, use_goal)
    end = np.broadcast_to(goal, (B, DOF)) if use_goal else u[:, -1]
    np.testing.assert_allclose(out["position"][:, -1], end, atol=1e-6)
    np.testing.assert_allclose(out["position"][:, -3], end, atol=1e-6)  # replicated tail
    np.testing.assert_allclose(out["velocity"][:, -1], 0.0, atol=1e-5)
    np.testing.assert_allclose(out["acceleration"][:, -1], 0.0, atol=1e-3)


def test_backward_is_transpose_of_forward_for_interior_actions(oracle):
    """<J du, g> = <du, J^T g> restricted to actions whose stencil windows are fully interior
    (the reference truncates the adjoint of the replicated tail, :300-352)"""
    rng = np.random.default_rng(2)
    u = rng.normal(size=(B, H - 4, DOF)).astype(np.float32)
    start = {k: np.zeros((1, DOF), np.float32) for k in ("position", "velocity", "acceleration")}
    goal = np.zeros((1, DOF), np.float32)
    g = [rng.normal(size=(B, H, DOF)).astype(np.float32) * s for s in (1.0, DT, DT ** 2, DT ** 3)]
    z = np.zeros(B, np.int32)
    gu = oracle.differentiation_position_backward(*g, np.array([DT], np.float32), z, np.zeros(1, np.uint8))
    base = _forward(oracle, u, start, goal, 0)
    keys = ("position", "velocity", "acceleration", "jerk")
    for ah in (0, 3, 9, H - 4 - 3):
        du = np.zeros_like(u)
        du[:, ah, :] = 1.0
        pert = _forward(oracle, u + du, start, goal, 0)
        lhs = sum(((pert[k].astype(np.float64) - base[k]) * gg).sum(axis=(1,)) for k, gg in zip(keys, g))  # [B, DOF]
        np.testing.assert_allclose(gu[:, ah, :], lhs, rtol=2e-3, atol=2e-3 * np.abs(lhs).max())


def test_backward_last_action_under_the_implicit_goal_is_the_true_transpose(oracle):
    """with the implicit goal the forward pass replaces the last action by the goal, so that action gets NO gradient
    (reference differentiation_position_kernel.cuh:352-361; the stencil expression next to it there is commented out) --
    checked as a transpose: perturbing the last action changes nothing"""
    rng = np.random.default_rng(4)
    u = rng.normal(size=(B, H - 4, DOF)).astype(np.float32)
    start = {k: (rng.normal(size=(1, DOF)) * 0.2).astype(np.float32) for k in ("position", "velocity", "acceleration")}
    goal = (rng.normal(size=(1, DOF)) * 0.2).astype(np.float32)
    g = [rng.normal(size=(B, H, DOF)).astype(np.float32) * s for s in (1.0, DT, DT ** 2, DT ** 3)]
    gu = oracle.differentiation_position_backward(*g, np.array([DT], np.float32), np.zeros(B, np.int32), np.ones(1, np.uint8))
    base = _forward(oracle, u, start, goal, 1)
    du = np.zeros_like(u)
    du[:, -1, :] = 0.7
    pert = _forward(oracle, u + du, start, goal, 1)
    for k in ("position", "velocity", "acceleration", "jerk"):
        assert np.array_equal(pert[k], base[k]), k
    assert np.all(gu[:, -1, :] == 0.0)


def test_backward_goal_state_masks_last_action_position_gradient(oracle):
    rng = np.random.default_rng(3)
    g = [rng.normal(size=(B, H, DOF)).astype(np.float32) for _ in range(4)]
    z = np.zeros(B, np.int32)
    dt = np.array([DT], np.float32)
    free = oracle.differentiation_position_backward(*g, dt, z, np.zeros(1, np.uint8))
    goal = oracle.differentiation_position_backward(*g, dt, z, np.ones(1, np.uint8))
    np.testing.assert_array_equal(free[:, :-1], goal[:, :-1])
    g2 = [x.copy() for x in g]
    g2[0][:] = 0.0  # position gradients must not reach the (goal-overridden) last action
    goal2 = oracle.differentiation_position_backward(*g2, dt, z, np.ones(1, np.uint8))
    np.testing.assert_array_equal(goal[:, -1], goal2[:, -1])


def test_integration_acceleration_is_semi_implicit_euler(oracle):
    rng = np.random.default_rng(4)
    u = rng.normal(size=(B, H, DOF)).astype(np.float32)
    start = {k: rng.normal(size=(2, DOF)).astype(np.float32) for k in ("position", "velocity", "acceleration")}
    sidx = np.array([0, 1, 1], np.int32)
    dt = rng.uniform(0.01, 0.1, size=H).astype(np.float32)
    out = oracle.integration_acceleration(u, start, sidx, dt)
    acc = np.concatenate([start["acceleration"][sidx][:, None], u[:, :-1]], axis=1).astype(np.float64)
    vel = start["velocity"][sidx][:, None] + np.cumsum(np.concatenate([np.zeros((B, 1, DOF)), acc[:, 1:] * dt[None, 1:, None]], 1), 1)
    pos = start["position"][sidx][:, None] + np.cumsum(np.concatenate([np.zeros((B, 1, DOF)), vel[:, 1:] * dt[None, 1:, None]], 1), 1)
    np.testing.assert_allclose(out["acceleration"], acc, atol=1e-6)
    np.testing.assert_allclose(out["velocity"], vel, atol=1e-5)
    np.testing.assert_allclose(out["position"], pos, atol=1e-5)
    jerk = np.zeros_like(acc)
    jerk[:, 1:] = (acc[:, 1:] - acc[:, :-1]) / dt[None, 1:, None]
    np.testing.assert_allclose(out["jerk"], jerk, rtol=1e-4, atol=1e-3)
