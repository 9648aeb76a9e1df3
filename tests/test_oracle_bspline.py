"""Pins for the B-spline oracle: boundary conditions the reference documents
(bspline_interpolation.cuh:110-124: start state fixes the first knots, the last knot is replicated
to come to rest, implicit goal state enforces the goal) and VJP vs finite differences
(reference curobo/tests/_src/transition/test_transition_gradients.py)."""

import numpy as np
import pytest


def _case(oracle, degree, implicit, rng, b=3, nk=10, dof=4, interp=3, dt=0.07):
    ph = (nk + degree + 1) * interp + 1
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda: {k: (rng.normal(size=(1, dof)) * s).astype(np.float32)  # noqa: E731
                  for k, s in (("position", 1.0), ("velocity", 0.3), ("acceleration", 0.2), ("jerk", 0.1))}
    start, goal = mk(), mk()
    idx = np.zeros(b, np.int32)
    out = oracle.bspline_forward(u, start, goal, idx, idx, np.array([dt], np.float32),
                                 np.array([implicit], np.uint8), ph, degree)
    return u, start, goal, out, ph, dt, interp


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_start_boundary_conditions(degree, oracle):
    u, start, goal, out, ph, dt, interp = _case(oracle, degree, False, np.random.default_rng(degree))
    np.testing.assert_allclose(out["position"][:, 0], np.broadcast_to(start["position"], out["position"][:, 0].shape), atol=2e-5)
    np.testing.assert_allclose(out["velocity"][:, 0], np.broadcast_to(start["velocity"], out["velocity"][:, 0].shape), atol=2e-4)
    np.testing.assert_allclose(out["acceleration"][:, 0], np.broadcast_to(start["acceleration"], out["acceleration"][:, 0].shape), atol=3e-3)
    if degree >= 4:  # a cubic cannot control jerk (FixedKnotCoeffsData<3> jerk row is zero)
        np.testing.assert_allclose(out["jerk"][:, 0], np.broadcast_to(start["jerk"], out["jerk"][:, 0].shape), atol=5e-2)
    np.testing.assert_allclose(out["dt"], dt)


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_replicated_goal_comes_to_rest_at_last_knot(degree, oracle):
    u, start, goal, out, ph, dt, interp = _case(oracle, degree, False, np.random.default_rng(10 + degree))
    np.testing.assert_allclose(out["position"][:, -1], u[:, -1], atol=1e-5)
    np.testing.assert_allclose(out["velocity"][:, -1], 0.0, atol=1e-4)
    np.testing.assert_allclose(out["acceleration"][:, -1], 0.0, atol=2e-3)


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_implicit_goal_state_is_reached(degree, oracle):
    """With an implicit goal the reference re-uses the START fixed-knot coefficients for the goal
    side (bspline_boundary_constraint.cuh:330-367), so the goal state is met at the beginning of
    the last knot interval (h = horizon - interpolation_steps); the remaining interval is the
    constant-acceleration continuation of that state."""
    u, start, goal, out, ph, dt, interp = _case(oracle, degree, True, np.random.default_rng(20 + degree))
    h = ph - 1 - interp
    bc = lambda a, ref: np.broadcast_to(ref, a.shape)  # noqa: E731
    np.testing.assert_allclose(out["position"][:, h], bc(out["position"][:, h], goal["position"]), atol=3e-5)
    np.testing.assert_allclose(out["velocity"][:, h], bc(out["velocity"][:, h], goal["velocity"]), atol=3e-4)
    np.testing.assert_allclose(out["acceleration"][:, h], bc(out["acceleration"][:, h], goal["acceleration"]), atol=5e-3)
    # the last free knot does not influence the trajectory (bspline_interpolation.cuh:186-205)
    u2 = u.copy()
    u2[:, -1] += 1.0
    idx = np.zeros(u.shape[0], np.int32)
    out2 = oracle.bspline_forward(u2, start, goal, idx, idx, np.array([dt], np.float32), np.array([1], np.uint8), ph, degree)
    np.testing.assert_array_equal(out2["position"], out["position"])


@pytest.mark.parametrize("degree", [3, 5])
def test_derivatives_are_consistent(degree, oracle):
    """velocity ~ d position / dt by central differences inside the spline"""
    u, start, goal, out, ph, dt, interp = _case(oracle, degree, False, np.random.default_rng(30 + degree), interp=8, dt=0.01)
    p, v = out["position"].astype(np.float64), out["velocity"].astype(np.float64)
    fd = (p[:, 2:] - p[:, :-2]) / (2 * dt)
    np.testing.assert_allclose(v[:, 1:-1][:, 5:-10], fd[:, 5:-10], atol=2e-2 * max(1.0, np.abs(fd).max()))


@pytest.mark.parametrize("degree", [3, 4, 5])
@pytest.mark.parametrize("implicit", [False, True])
def test_backward_is_the_transpose_of_forward(degree, implicit, oracle):
    """The map knots -> (p, v, a, j) is affine, so <J u, g> == <u, J^T g> exactly (up to fp32)
    once the constant (start/goal) part is removed; this checks the VJP against the forward
    oracle without finite-difference noise."""
    rng = np.random.default_rng(40 + degree + int(implicit))
    b, nk, dof, interp, dt = 2, 9, 3, 2, 0.05
    ph = (nk + degree + 1) * interp + 1
    z = {k: np.zeros((1, dof), np.float32) for k in ("position", "velocity", "acceleration", "jerk")}
    idx = np.zeros(b, np.int32)
    dts, imp = np.array([dt], np.float32), np.array([implicit], np.uint8)
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    f = oracle.bspline_forward(u, z, z, idx, idx, dts, imp, ph, degree)  # linear part only (zero boundary states)
    g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
    # scale derivative gradients so all four terms contribute comparably
    g[1] *= dt * interp
    g[2] *= (dt * interp) ** 2
    g[3] *= (dt * interp) ** 3
    gu = oracle.bspline_backward(*g, dts, idx, imp, nk, degree)
    lhs = sum((f[k].astype(np.float64) * gg).sum() for k, gg in zip(("position", "velocity", "acceleration", "jerk"), g))
    rhs = (u.astype(np.float64) * gu).sum()
    assert lhs == pytest.approx(rhs, rel=2e-4, abs=1e-3)
    if implicit:  # last knot is ignored with an implicit goal state: zero gradient
        assert not gu[:, -1].any()


@pytest.mark.parametrize("degree", [3, 5])
def test_single_dt_reinterpolation_matches_forward_per_trajectory(degree, oracle):
    """The single-dt kernel is the forward interpolation with interpolation_dt[0] and a
    per-trajectory horizon (bspline_kernel.cuh:221-270): every trajectory must equal the plain
    forward kernel run at its own horizon, and points past its horizon repeat the last sample."""
    rng = np.random.default_rng(degree)
    b, nk, dof, max_out = 4, 8, 3, 80
    sup = degree + 1
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda n: {k: (rng.normal(size=(n, dof)) * 0.2).astype(np.float32)  # noqa: E731
                    for k in ("position", "velocity", "acceleration", "jerk")}
    start, goal = mk(2), mk(2)
    sidx = np.array([0, 1, 1, 0], np.int32)
    gidx = np.array([1, 0, 1, 0], np.int32)
    horizons = np.array([(nk + sup) * 2, (nk + sup) * 4, (nk + sup) * 3, 500], np.int32)  # last one is clamped
    dt = np.array([0.03], np.float32)
    imp = np.zeros(2, np.uint8)
    out = oracle.bspline_single_dt(u, start, goal, sidx, gidx, dt, imp, horizons, max_out, degree)
    np.testing.assert_allclose(out["dt"], 0.03)
    for i in range(b):
        nh = min(int(horizons[i]), max_out - 1)
        ref = oracle.bspline_forward(u[i:i + 1], start, goal, sidx[i:i + 1], gidx[i:i + 1], np.array([0.03, 0.03], np.float32),
                                     imp, nh + 1, degree)
        for k in ("position", "velocity", "acceleration", "jerk"):
            np.testing.assert_array_equal(out[k][i, :nh + 1], ref[k][0])
            if nh + 1 < max_out:  # clamped tail = the sample at t = 1 of the last knot interval
                np.testing.assert_array_equal(out[k][i, nh + 1:], np.broadcast_to(out[k][i, nh + 1], out[k][i, nh + 1:].shape))
        if nh + 1 < max_out and (nh % (nk + sup)) == 0:
            np.testing.assert_allclose(out["position"][i, nh + 1], out["position"][i, nh], atol=1e-5)


def test_retiming_helpers_match_reference_golden():
    """calculate_dt_no_clamp / calculate_traj_steps (curobo_amd/util/trajectory.py) against the outputs of the
    reference's own functions (tests/golden/make_retime_golden.py)"""
    import os

    import torch

    from curobo_amd.util.trajectory import calculate_dt_no_clamp, calculate_traj_steps

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "retime_golden.npz"))
    t = lambda k: torch.as_tensor(g[k])  # noqa: E731
    score = calculate_dt_no_clamp(t("vel"), t("acc"), t("jerk"), t("max_vel"), t("max_acc"), t("max_jerk"), epsilon=1e-3)
    np.testing.assert_allclose(score.numpy(), g["score"], rtol=1e-6)
    for ni in (0, 1):
        steps, smax = calculate_traj_steps(t("dt"), t("idt"), 17, nearest_int=bool(ni))
        assert np.array_equal(steps.numpy(), g[f"steps_{ni}"]) and int(smax) == int(g[f"steps_max_{ni}"])
    # scaling property: a trajectory slowed down by the score meets its tightest limit exactly
    s = score.view(-1, 1, 1) / (1.0 + 1e-3)
    worst = torch.maximum(torch.maximum((t("vel") / s).abs().amax(1) / t("max_vel"), (t("acc") / s ** 2).abs().amax(1) / t("max_acc")),
                          (t("jerk") / s ** 3).abs().amax(1) / t("max_jerk")).amax(-1)
    np.testing.assert_allclose(worst.numpy(), 1.0, rtol=1e-5)


def test_trajectory_seed_generator_matches_reference_golden():
    """curobo_amd.util.knot_seeds == the reference's TrajectorySeedGenerator
    (tests/golden/make_trajectory_seed_golden.py)"""
    import os

    import torch

    from conftest import GOLDEN_DIR
    from curobo_amd.util.knot_seeds import TrajectorySeedGenerator

    g = np.load(os.path.join(GOLDEN_DIR, "trajectory_seed_golden.npz"))
    B, S, H, D = g["interpolated"].shape
    gen = TrajectorySeedGenerator(H, D)
    out = gen.generate_interpolated_seeds(torch.as_tensor(g["start"]), torch.as_tensor(g["goal"]), S)
    np.testing.assert_array_equal(out.numpy(), g["interpolated"])
    np.testing.assert_array_equal(gen.generate_constant_seeds(torch.as_tensor(g["start"]), S).numpy(), g["constant"])
    np.testing.assert_array_equal(out[:, :, 0].numpy(), np.broadcast_to(g["start"][:, None], (B, S, D)))
    np.testing.assert_allclose(out[:, :, -1].numpy(), g["goal"], atol=1e-7)
    with pytest.raises(ValueError):
        gen.generate_interpolated_seeds(torch.as_tensor(g["start"]), torch.as_tensor(g["goal"][:, :2]), S)


# ---------------------------------------------------------------------------------------------
# Pins from outside the repository (tests/golden/make_bspline_golden.py): the reference's own
# derivation script for the fixed-knot coefficients + scipy.interpolate.BSpline for the basis.
# ---------------------------------------------------------------------------------------------
def _bspline_golden():
    import os

    from conftest import GOLDEN_DIR

    return np.load(os.path.join(GOLDEN_DIR, "bspline_golden.npz"))


def _table_from_source(path, name_regex, rows, cols):
    """the float table `name` of a C / HIP source as an array (entries are literals like -3.0f / 2.0f)"""
    import re

    text = open(path).read()
    m = re.search(name_regex + r"[^=]*=\s*\{(.*?)\};", text, flags=re.S)
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", m.group(1))
    vals = [eval(tok.replace("f", "")) for tok in re.findall(r"-?[0-9.]+f(?:\s*/\s*[0-9.]+f)?", body)]  # noqa: S307
    return np.array(vals, np.float64).reshape(rows, cols)


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_fixed_knot_tables_equal_the_reference_derivation(degree):
    """kFix3/4/5 (HIP) and c3/c4/c5 (oracle) == the coefficients the reference's derivation script
    produces (bspline_boundary_coefficients.py, run by the golden generator); the cubic's jerk row is
    zero in the reference table (bspline_boundary_constraint.cuh:61) although derivable."""
    import os

    from conftest import ROOT

    gold = _bspline_golden()
    want = gold[f"coeffs_table_{degree}"]
    hip = _table_from_source(os.path.join(ROOT, "curobo_amd", "csrc", "bspline_device.hpp"), rf"kFix{degree}\[4\]\[{degree + 1}\]", 4, degree + 1)
    orc = _table_from_source(os.path.join(ROOT, "oracle", "curobo_oracle.c"), rf"c{degree}\[4\]\[{degree + 1}\]", 4, degree + 1)
    np.testing.assert_allclose(hip, want, atol=5e-7)  # -0.833333f is the reference's own truncation of -5/6
    np.testing.assert_allclose(orc, want, atol=5e-7)
    if degree == 3:
        np.testing.assert_allclose(gold["coeffs_3"][3], [0, 0, 0, 1], atol=1e-12)


@pytest.mark.parametrize("implicit", [0, 1])
@pytest.mark.parametrize("degree", [3, 4, 5])
def test_whole_trajectory_matches_scipy_bspline(degree, implicit, oracle):
    """position / velocity / acceleration / jerk at EVERY sample vs scipy.interpolate.BSpline on the
    control sequence [fixed start knots | free knots | replicated last knot or fixed goal knots]
    (float64; fixed knots from the reference-derived coefficients)."""
    gold = _bspline_golden()
    key = f"d{degree}_g{implicit}"
    n, dof, interp, dt, H = gold[key + "_meta"]
    n, dof, interp, H = int(n), int(dof), int(interp), int(H)
    u = gold[key + "_u"].astype(np.float32)
    names = ("position", "velocity", "acceleration", "jerk")
    start = {k: gold[key + "_start"][i][None].astype(np.float32) for i, k in enumerate(names)}
    goal = {k: gold[key + "_goal"][i][None].astype(np.float32) for i, k in enumerate(names)}
    idx = np.zeros(u.shape[0], np.int32)
    out = oracle.bspline_forward(u, start, goal, idx, idx, np.array([dt], np.float32), np.array([implicit], np.uint8), H, degree)
    want = gold[key + "_out"]
    for i, k in enumerate(names):
        scale = max(1.0, np.abs(want[i]).max())
        np.testing.assert_allclose(out[k], want[i], atol=2e-5 * scale * (10.0 ** i), rtol=0, err_msg=k)


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_basis_functions_match_scipy(degree, oracle):
    """interior segments (no boundary knots involved): samples = sum_i u[seg - support + i] N_i(t) with
    scipy's N_i and its derivatives"""
    gold = _bspline_golden()
    basis, ts = gold[f"basis_{degree}"], gold["basis_t"]
    sup, interp, dt = degree + 1, 8, 0.05
    n = 3 * sup
    rng = np.random.default_rng(degree)
    u = rng.normal(size=(1, n, 2)).astype(np.float32)
    z = {k: np.zeros((1, 2), np.float32) for k in ("position", "velocity", "acceleration", "jerk")}
    idx = np.zeros(1, np.int32)
    H = (n + sup) * interp + 1
    out = oracle.bspline_forward(u, z, z, idx, idx, np.array([dt], np.float32), np.array([0], np.uint8), H, degree)
    knot_dt = dt * interp
    for seg in range(sup, n):  # segments whose support lies inside the free knots
        for ti in range(interp):
            h = seg * interp + ti
            assert abs(ts[ti] - ti / interp) < 1e-12
            ctrl = u[0, seg - sup:seg].astype(np.float64)  # [support, dof]
            for der, k in enumerate(("position", "velocity", "acceleration", "jerk")):
                want = basis[der, ti] @ ctrl / knot_dt ** der
                np.testing.assert_allclose(out[k][0, h], want, atol=1e-5 * max(1.0, np.abs(want).max()) * 3.0 ** der)


def test_randomised_sweep_of_the_host_utilities_against_the_reference():
    """tests/randomised/sweep_reference_torch_util.py at a small size (skips itself where /root/reference is absent)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", "sweep_reference_torch_util.py"), "60", "9"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout or "0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]
