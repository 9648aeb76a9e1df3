"""The RNEA oracle (oracle/curobo_oracle.c, A8) against golden vectors computed by the reference's
own NumPy implementation (tests/golden/make_rnea_golden.py imports
curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py), plus the properties the reference's
test suite checks (test_rnea_reference.py: gravity-only torques, VJP vs finite differences)."""

import os

import numpy as np
import pytest

from conftest import load_model

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rnea_golden.npz")


@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_rnea_forward_matches_reference_numpy(robot, oracle):
    g = np.load(GOLD)
    m = load_model(robot).as_dict()
    tau, cache = oracle.rnea_forward(g[f"{robot}/q"], g[f"{robot}/qd"], g[f"{robot}/qdd"], m)
    scale = max(1.0, np.abs(g[f"{robot}/tau"]).max())
    np.testing.assert_allclose(tau, g[f"{robot}/tau"], rtol=2e-4, atol=2e-5 * scale)
    for name, sl in (("v", slice(0, 6)), ("a", slice(6, 12)), ("f", slice(12, 18))):
        ref = g[f"{robot}/{name}"]
        np.testing.assert_allclose(cache[:, :, sl], ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
    assert np.all(cache[:, :, 18:] == 0.0)


@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_rnea_backward_matches_reference_numpy(robot, oracle):
    g = np.load(GOLD)
    m = load_model(robot).as_dict()
    _, cache = oracle.rnea_forward(g[f"{robot}/q"], g[f"{robot}/qd"], g[f"{robot}/qdd"], m)
    gq, gqd, gqdd = oracle.rnea_backward(g[f"{robot}/tau_bar"], g[f"{robot}/q"], g[f"{robot}/qd"], cache, m)
    for ours, name in ((gq, "grad_q"), (gqd, "grad_qd"), (gqdd, "grad_qdd")):
        ref = g[f"{robot}/{name}"]
        np.testing.assert_allclose(ours, ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_rnea_static_torques_are_gravity_torques(oracle):
    """qd = qdd = 0: tau is the gravity torque; it vanishes without gravity and flips with it"""
    m = load_model("franka").as_dict()
    rng = np.random.default_rng(0)
    lo, hi = m["joint_limits_position"]
    q = rng.uniform(lo, hi, size=(5, 7)).astype(np.float32)
    z = np.zeros_like(q)
    tau_g, _ = oracle.rnea_forward(q, z, z, m)
    tau_0, _ = oracle.rnea_forward(q, z, z, m, gravity=(0, 0, 0, 0, 0, 0))
    tau_n, _ = oracle.rnea_forward(q, z, z, m, gravity=(0, 0, 0, 0, 0, -9.81))
    assert np.abs(tau_g).max() > 1.0
    np.testing.assert_allclose(tau_0, 0.0, atol=1e-6)
    np.testing.assert_allclose(tau_n, -tau_g, atol=1e-5)
    # the base joint axis is vertical: gravity produces no torque about it
    np.testing.assert_allclose(tau_g[:, 0], 0.0, atol=1e-5)


def test_rnea_vjp_matches_finite_differences(oracle):
    m = load_model("franka").as_dict()
    rng = np.random.default_rng(1)
    lo, hi = m["joint_limits_position"]
    q = rng.uniform(lo, hi, size=(2, 7))
    qd, qdd = rng.normal(size=(2, 7)), rng.normal(size=(2, 7))
    w = rng.normal(size=(2, 7))
    _, cache = oracle.rnea_forward(q, qd, qdd, m)
    grads = oracle.rnea_backward(w, q, qd, cache, m)
    eps = 2e-3
    for gi, x in enumerate((q, qd, qdd)):
        fd = np.zeros((2, 7))
        for j in range(7):
            args_p = [q.copy(), qd.copy(), qdd.copy()]
            args_m = [q.copy(), qd.copy(), qdd.copy()]
            args_p[gi][:, j] += eps
            args_m[gi][:, j] -= eps
            tp, _ = oracle.rnea_forward(*args_p, m)
            tm, _ = oracle.rnea_forward(*args_m, m)
            fd[:, j] = ((tp.astype(np.float64) - tm) * w).sum(1) / (2 * eps)
        np.testing.assert_allclose(grads[gi], fd, rtol=2e-2, atol=2e-2 * np.abs(fd).max())


def test_rnea_external_force_enters_linearly_with_negative_sign(oracle):
    m = load_model("franka").as_dict()
    rng = np.random.default_rng(2)
    lo, hi = m["joint_limits_position"]
    q = rng.uniform(lo, hi, size=(3, 7))
    qd, qdd = rng.normal(size=(3, 7)), rng.normal(size=(3, 7))
    L = m["fixed_transforms"].shape[0]
    fe = rng.normal(size=(3, L, 6)).astype(np.float32)
    t0, _ = oracle.rnea_forward(q, qd, qdd, m)
    t1, c1 = oracle.rnea_forward(q, qd, qdd, m, f_ext=fe)
    t2, _ = oracle.rnea_forward(q, qd, qdd, m, f_ext=2 * fe)
    np.testing.assert_allclose(t2 - t0, 2 * (t1 - t0), rtol=1e-3, atol=1e-4)
    w = rng.normal(size=(3, 7))
    *_, gfe = oracle.rnea_backward(w, q, qd, c1, m, want_f_ext_grad=True)
    # d<w, tau>/d f_ext by finite differences on one entry per link
    for k in (0, 5, L - 1):
        d = np.zeros_like(fe)
        d[:, k, 2] = 1e-2
        tp, _ = oracle.rnea_forward(q, qd, qdd, m, f_ext=fe + d)
        fd = ((tp - t1) * w).sum(1) / 1e-2
        np.testing.assert_allclose(gfe[:, k, 2], fd, rtol=2e-2, atol=2e-3)


@pytest.mark.skipif(not os.path.isfile("/root/reference/curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py"),
                    reason="the reference's NumPy RNEA is not on this machine")
def test_randomised_sweep_against_the_reference_numpy_rnea():
    """tests/randomised/sweep_reference_numpy_rnea.py at a small size: random configurations, velocity / acceleration scales, robots"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", "sweep_reference_numpy_rnea.py"), "40", "9"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]
