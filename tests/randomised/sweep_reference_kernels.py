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
        # L-BFGS step kernels (both variants) over a few iterations of one state, with degenerate pairs: no movement (y = s = 0),
        # negative curvature, zero gradients, tiny / huge scales; stable mode on and off.  History buffers identical, rho and the
        # step to the rounding of the block reductions, the same finite pattern
        ob, ov, om = int(rng.integers(1, 6)), int(rng.choice([7, 33, 84, 175])), int(rng.choice([1, 5, 15]))
        om = min(om, ov)  # (the reference launches v threads and moves its rho buffer with `threadIdx.x < history`: for v < history
        #                   the kernel is not the algorithm any more -- tests/test_reference_cuda_kernels.py documents that case)
        stable, shared = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        zf = lambda *sh: np.zeros(sh, np.float32)  # noqa: E731
        A = dict(step=zf(ob, ov), rho=zf(om, ob), y=zf(om, ob, ov), s=zf(om, ob, ov), x0=zf(ob, ov), g0=zf(ob, ov))
        Bk = {k: a.copy() for k, a in A.items()}
        x = rng.normal(size=(ob, ov)).astype(np.float32)
        gscale = np.float32(10.0 ** rng.integers(-3, 4))  # (beyond 1e3 a degenerate pair in the history leaves no digits to compare)
        last_deg = -10 ** 6
        for it in range(om + 2):
            mode = int(rng.integers(0, 5))
            if mode == 2 and not stable:
                mode = 0  # (without the stable-mode guards a negative-curvature pair makes the recursion indefinite: nothing to compare)
            if mode in (1, 2, 3):
                last_deg = it
            # a degenerate pair inside the history makes the two-loop recursion a cancellation of large terms: the kernels' tree
            # reductions and the oracle's index-order sums then differ by rounding x the condition number (transient: measured up to
            # a few percent of the step for one iteration, more under gradient scales of 1e4 .. 1e6); a SEMANTIC difference would be O(1)
            # on every element and stay
            loose = it - last_deg <= om
            if mode != 1:
                x = (x + 0.05 * rng.normal(size=(ob, ov))).astype(np.float32)  # (mode 1: no movement since the last iteration)
            g = ((2.0 * x + 0.1 * rng.normal(size=(ob, ov))) * gscale).astype(np.float32)
            if mode == 1 and it > 0:
                g = g_prev.copy()
            if mode == 2 and it > 0:
                g[0] = A["g0"][0] - (x[0] - A["x0"][0]) * gscale  # y . s < 0 on one problem
            if mode == 3:
                g[rng.random(ob) < 0.5] = 0
            g_prev = g
            oracle.lbfgs_step(A["step"], A["rho"], A["y"], A["s"], x, g, A["x0"], A["g0"], 0.01, stable)
            ref.lbfgs_step(Bk["step"], Bk["rho"], Bk["y"], Bk["s"], x, g, Bk["x0"], Bk["g0"], 0.01, stable, shared_buffers=shared)
            for k in ("y", "s", "x0", "g0"):
                assert np.array_equal(A[k], Bk[k]), f"L-BFGS {k} (iteration {it})"
            fr = np.isfinite(A["rho"])
            assert np.array_equal(fr, np.isfinite(Bk["rho"])), f"L-BFGS rho finite pattern (iteration {it}, mode {mode}, stable {stable})"
            np.testing.assert_allclose(Bk["rho"][fr], A["rho"][fr], rtol=5e-3 if loose else 5e-5,  # (y . s of a degenerate pair is a cancellation)
                                       atol=1e-7 * max(1e-30, float(np.abs(A["rho"][fr]).max()) if fr.any() else 1.0),
                                       err_msg=f"L-BFGS rho (iteration {it}, mode {mode}, stable {stable}, shared {shared}, b {ob} v {ov} m {om} gscale {gscale})")
            fin = np.isfinite(A["step"])
            assert np.array_equal(fin, np.isfinite(Bk["step"])), f"L-BFGS step finite pattern (iteration {it}, mode {mode}, stable {stable}, shared {shared})"
            if fin.any() and (stable or not loose):  # (without the stable-mode guards a degenerate pair leaves inf / huge rho in the history:
                #                                        only the finite pattern is comparable)
                # per problem: a history with a near-singular pair amplifies the reductions' rounding
                for r in range(ob):
                    fr_ = fin[r]
                    if fr_.any():
                        sc = float(np.abs(A["step"][r][fr_]).max())
                        np.testing.assert_allclose(Bk["step"][r][fr_], A["step"][r][fr_], rtol=5e-3, atol=(3e-1 if loose else 2e-4) * max(sc, 1e-30),
                                                   err_msg=f"L-BFGS step (iteration {it}, mode {mode}, stable {stable}, shared {shared}, b {ob} v {ov} m {om} gscale {gscale}, row scale {sc:.3e})")
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: {robot} n {n} scale {scale}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
