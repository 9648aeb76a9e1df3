"""The oracle against the reference's own CUDA kernels run on the CPU (oracle/_ref/libcurobo_ref.so) on RANDOM inputs: batch
sizes, joint ranges beyond the limits, sparse gradients, spline shapes.  FK, self collision and the B-spline kernels must
agree to the last bit; the FK VJP and RNEA to summation rounding.  CPU only.
    python tests/randomised/sweep_reference_kernels.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402

from oracle import ref_kernels  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

if not ref_kernels.available():
    print("oracle/_ref/libcurobo_ref.so is not built here: nothing to compare; 0 failed")
    sys.exit(0)
oracle, ref = Oracle(), ref_kernels.ReferenceKernels()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KEYS = ("position", "velocity", "acceleration", "jerk")
models = {r: load_model(r) for r in ("franka", "ur10e", "unitree_g1")}
bad = 0
for case in range(n_cases):
    robot = str(rng.choice(list(models), p=[0.45, 0.45, 0.1]))
    model = models[robot]
    md = model.as_dict()
    n = int(rng.integers(1, 24 if robot != "unitree_g1" else 5))
    scale = float(rng.choice([0.3, 1.0, 1.5, 6.0]))
    q = (sample_q(model, n, seed=int(rng.integers(10000))) * scale).astype(np.float32)
    try:
        a = oracle.kinematics_forward(q, md, compute_jacobian=True, compute_com=True)
        b = ref.kinematics_forward(q, md, compute_jacobian=True, compute_com=True)
        for k in ("link_pos", "link_quat", "cumul_mat", "robot_spheres", "jacobian"):
            assert np.array_equal(a[k], b[k]), f"FK {k} not bit-identical"
        S, T = a["robot_spheres"].shape[1], a["link_pos"].shape[1]
        gs = rng.standard_normal((n, S, 4)).astype(np.float32)
        gs[..., 3] = 0
        gs[rng.random((n, S)) < float(rng.choice([0.0, 0.5, 0.97]))] = 0
        gp, gq = rng.standard_normal((n, T, 3)).astype(np.float32), rng.standard_normal((n, T, 4)).astype(np.float32)
        va, vb = oracle.kinematics_backward(md, a["cumul_mat"], gs, gp, gq), ref.kinematics_backward(md, a["cumul_mat"], gs, gp, gq)
        np.testing.assert_allclose(vb, va, rtol=0, atol=3e-6 * max(np.abs(va).max(), 1e-6), err_msg="FK VJP")
        if robot != "unitree_g1":
            sa = oracle.self_collision(a["robot_spheres"], model.sphere_padding, model.collision_pairs, 1.5)
            sb = ref.self_collision(a["robot_spheres"], model.sphere_padding, model.collision_pairs, 1.5)
            for k in ("distance", "gradient", "sparse_index"):
                assert np.array_equal(sa[k], sb[k]), f"self collision {k} not bit-identical"
        if robot != "ur10e":
            qd, qdd, gt = (rng.standard_normal(q.shape).astype(np.float32) * float(rng.choice([0.1, 1.0, 8.0])) for _ in range(3))
            ta, ca = oracle.rnea_forward(q, qd, qdd, md)
            tb, cb = ref.rnea_forward(q, qd, qdd, md)
            np.testing.assert_allclose(tb, ta, rtol=0, atol=2e-6 * max(np.abs(ta).max(), 1e-6), err_msg="RNEA tau")
            for x, y in zip(oracle.rnea_backward(gt, q, qd, ca, md), ref.rnea_backward(gt, q, qd, cb, md)):
                np.testing.assert_allclose(y, x, rtol=0, atol=5e-6 * max(np.abs(x).max(), 1e-6), err_msg="RNEA VJP")
        # B-spline kernels
        degree, nk, dof, interp, bb = int(rng.choice([3, 4, 5])), int(rng.choice([2, 6, 12, 20])), int(rng.choice([1, 6, 7, 12])), int(rng.choice([1, 2, 4])), int(rng.integers(1, 12))
        ph = (nk + degree + 1) * interp + 1
        u = rng.normal(size=(bb, nk, dof)).astype(np.float32)
        ns, ng = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        mk = lambda m: {k: (rng.normal(size=(m, dof)) * 0.3).astype(np.float32) for k in KEYS}  # noqa: E731
        start, goal = mk(ns), mk(ng)
        sidx, gidx = rng.integers(0, ns, size=bb).astype(np.int32), rng.integers(0, ng, size=bb).astype(np.int32)
        dt, imp = rng.uniform(0.01, 0.2, size=ng).astype(np.float32), np.full(ng, int(rng.integers(2)), np.uint8)
        fa, fb = oracle.bspline_forward(u, start, goal, sidx, gidx, dt, imp, ph, degree), ref.bspline_forward(u, start, goal, sidx, gidx, dt, imp, ph, degree)
        for k in KEYS + ("dt",):
            assert np.array_equal(fa[k], fb[k]), f"B-spline forward {k} not bit-identical (degree {degree}, knots {nk}, dof {dof}, interp {interp})"
        g = [rng.normal(size=(bb, ph, dof)).astype(np.float32) for _ in range(4)]
        assert np.array_equal(oracle.bspline_backward(*g, dt, gidx, imp, nk, degree), ref.bspline_backward(*g, dt, gidx, imp, nk, degree)), "B-spline VJP"
        # Wolfe line search: random rounds with ties between candidates, flat costs and zero directions; every state array identical
        lb, nls, lv = int(rng.integers(1, 40)), 4, int(rng.choice([7, 84, 96]))
        z = np.zeros
        mkst = lambda: dict(best_cost=np.full((lb,), 1e9, np.float32), best_action=z((lb, lv), np.float32), best_iteration=z((lb,), np.int16),  # noqa: E731
                            current_iteration=z((lb,), np.int16), converged=z((lb,), np.uint8), exploration_cost=z((lb,), np.float32),
                            exploration_action=z((lb, lv), np.float32), exploration_gradient=z((lb, lv), np.float32), cost=z((lb,), np.float32),
                            action=z((lb, lv), np.float32), gradient=z((lb, lv), np.float32), exploration_idx=z((lb, nls), np.int32),
                            selected_idx=z((lb, nls), np.int32))
        sa, sb = mkst(), mkst()
        strong, approx = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
        if strong:
            approx = False
        for rnd in range(3):
            x, dd = rng.normal(size=(lb, nls, lv)).astype(np.float32), rng.normal(size=(lb, lv)).astype(np.float32)
            c = (rng.random((lb, nls)) * np.array([1, 0.8, 1.2, 2.0])).astype(np.float32)
            mode = int(rng.integers(0, 4))
            if mode == 1:
                c[:, 2] = c[:, 1]  # a tie between two candidates
            if mode == 2:
                c[rng.random(lb) < 0.5] = 0.5  # all candidates of a problem cost the same
            if mode == 3:
                dd[rng.random(lb) < 0.5] = 0  # zero direction
            gx = (rng.normal(size=(lb, nls, lv)) * 0.3).astype(np.float32)
            al = np.array([0.0, 0.25, 0.5, 1.0], np.float32)
            for impl, st in ((oracle, sa), (ref, sb)):
                impl.line_search(st, c, x, gx, dd, al, 1e-5, 0.9, strong, approx, 5, 0.0, 0.001)
            for k in sa:
                assert np.array_equal(sa[k], sb[k]), f"line search {k} (round {rnd}, mode {mode}, strong {strong}, approx {approx})"
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: {robot} n {n} scale {scale}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
