"""Golden vectors that pin the B-spline path against two sources outside this repository:

1. the REFERENCE's own derivation script of the boundary ("fixed knot") coefficients,
   curobo/_src/curobolib/kernels/trajectory/bspline/derivations/bspline_boundary_coefficients.py,
   imported and run here (its quartic / quintic functions print their result: stdout is parsed; for the
   cubic its `compute_cubic_bspline_derivatives(t=0)` rows are inverted exactly as the quartic / quintic
   functions do, which is the form the CUDA table bspline_boundary_constraint.cuh:52-92 holds -- the
   script's `derive_fixed_knot_coefficients_degree3` evaluates the same system one knot later, t = 1);
2. scipy.interpolate.BSpline (uniform knots) for the basis functions and their derivatives: whole
   trajectories (position, velocity, acceleration, jerk at every sample) of random knot sets, built from
   the control sequence [fixed start knots | free knots | replicated last knot or fixed goal knots] with
   the fixed knots computed from the coefficients of (1), in float64.

    python tests/golden/make_bspline_golden.py       (needs /root/reference; numpy + scipy only)
"""
import contextlib
import importlib.util
import io
import os
import re

import numpy as np
from scipy.interpolate import BSpline

REF = "/root/reference/curobo/_src/curobolib/kernels/trajectory/bspline/derivations/bspline_boundary_coefficients.py"
spec = importlib.util.spec_from_file_location("ref_bspline_derivation", REF)
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)


def parse_printed(fn):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn()
    text = buf.getvalue().split("Derived Fixed Knot Coefficients")[1]
    rows = []
    for name in ("Position", "Velocity", "Acceleration", "Jerk"):
        m = re.search(name + r" coefficients:\s*\[([^\]]*)\]", text, flags=re.S)
        rows.append(np.array([float(v) for v in m.group(1).split()]))
    return np.stack(rows)  # [4 (pos, vel, acc, jerk), degree + 1]


def cubic_at_t0():
    d = R.compute_cubic_bspline_derivatives(0.0, 1.0)
    M = np.stack([d["position"], d["velocity"], d["acceleration"], d["jerk"]])
    Minv = np.linalg.inv(M)
    return np.stack([Minv[:, 0], Minv[:, 1], Minv[:, 2], Minv[:, 3]])


coeffs = {3: cubic_at_t0(), 4: parse_printed(R.derive_fixed_knot_coefficients_degree4),
          5: parse_printed(R.derive_fixed_knot_coefficients_degree5)}
# the CUDA table zeroes the cubic's jerk row ("cubic can't control jerk", bspline_boundary_constraint.cuh:61)
coeffs_table = {k: v.copy() for k, v in coeffs.items()}
coeffs_table[3][3] = 0.0


def fixed_knots(C, state, knot_dt):
    """[support] control points that realise (pos, vel, acc, jerk) -- bspline_boundary_constraint.cuh:330-367"""
    p, v, a, j = state
    return C[0] * p + C[1] * v * knot_dt + C[2] * a * knot_dt ** 2 + C[3] * j * knot_dt ** 3


def trajectory(degree, u, start, goal, implicit, interp, dt):
    """float64 reference of interpolate_bspline_kernel for one dof: u [n_knots]; returns [4, padded_horizon]"""
    sup, n = degree + 1, u.shape[0]
    knot_dt = dt * interp
    C = coeffs_table[degree]
    ctrl = list(fixed_knots(C, start, knot_dt)) + list(u)
    if implicit:
        ctrl = ctrl[:-1] + list(fixed_knots(C, goal, knot_dt))  # the last free knot is replaced by the first goal knot
        ctrl = ctrl + [ctrl[-1]] * 2
    else:
        ctrl = ctrl + [u[-1]] * (sup + 1)
    ctrl = np.asarray(ctrl, np.float64)
    spl = BSpline(np.arange(len(ctrl) + degree + 1, dtype=np.float64), ctrl, degree)
    padded_n = n + sup
    H = padded_n * interp
    out = np.zeros((4, H + 1))
    for h in range(H + 1):
        seg, t = divmod(h, interp)
        t = t / interp
        if seg >= padded_n:
            seg, t = padded_n - 1, 1.0
        x = seg + degree + t
        # evaluate on the segment's own polynomial piece (t = 1 belongs to segment `seg`, not the next one)
        xe = min(x, seg + degree + 1 - 1e-12)
        for der in range(4):
            out[der, h] = (spl.derivative(der)(xe) if der else spl(xe)) / knot_dt ** der if der <= degree else 0.0
    return out


rng = np.random.default_rng(7)
gold = {}
for degree in (3, 4, 5):
    gold[f"coeffs_{degree}"] = coeffs[degree]
    gold[f"coeffs_table_{degree}"] = coeffs_table[degree]
    # basis functions and derivatives on a segment from scipy: N_i(p + t), i = 0..p
    ts = np.linspace(0.0, 1.0, 9)
    basis = np.zeros((4, len(ts), degree + 1))
    for i in range(degree + 1):
        c = np.zeros(2 * degree + 1)
        c[i] = 1.0
        b = BSpline(np.arange(len(c) + degree + 1, dtype=np.float64), c, degree)
        for der in range(4):
            x = np.minimum(degree + ts, degree + 1 - 1e-12)
            basis[der, :, i] = b.derivative(der)(x) if der else b(x)
    gold[f"basis_{degree}"] = basis
    gold["basis_t"] = ts
    for implicit in (0, 1):
        n, dof, interp, dt, B = 9, 3, 3, 0.07, 2
        u = rng.normal(size=(B, n, dof))
        st = [rng.normal(size=dof) * s for s in (1.0, 0.3, 0.2, 0.1)]
        go = [rng.normal(size=dof) * s for s in (1.0, 0.3, 0.2, 0.1)]
        H = (n + degree + 1) * interp + 1
        out = np.zeros((4, B, H, dof))
        for b in range(B):
            for d in range(dof):
                out[:, b, :, d] = trajectory(degree, u[b, :, d], [s[d] for s in st], [g[d] for g in go], bool(implicit), interp, dt)
        key = f"d{degree}_g{implicit}"
        gold[key + "_u"], gold[key + "_start"], gold[key + "_goal"] = u, np.stack(st), np.stack(go)
        gold[key + "_out"] = out
        gold[key + "_meta"] = np.array([n, dof, interp, dt, H])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bspline_golden.npz")
np.savez_compressed(path, **gold)
print(path, os.path.getsize(path))
for k in (3, 4, 5):
    print(k, np.round(coeffs[k], 6).tolist())
