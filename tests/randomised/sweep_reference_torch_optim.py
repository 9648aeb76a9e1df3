"""The oracle's L-BFGS step and Wolfe line search against the reference's OWN torch twins (lbfgs_jit_helpers.py,
line_search_strategy.py, imported from /root/reference as tests/golden/make_optim_golden.py does) on random shapes: batch,
optimisation dimension, history length, number of iterations, the three line-search strategies.  CPU only.
    python tests/randomised/sweep_reference_torch_optim.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/optim"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_optim_golden as G  # noqa: E402  (puts /root/reference on the path; runs the reference's torch code)

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    b, v, m, iters = int(rng.integers(1, 12)), int(rng.choice([1, 3, 7, 14, 84, 100])), int(rng.choice([1, 2, 7, 15, 27])), int(rng.integers(1, 9))
    try:
        g = G.lbfgs_case(rng, b, v, m, iters=iters)
        y, s, rho = np.zeros((m, b, v), np.float32), np.zeros((m, b, v), np.float32), np.zeros((m, b), np.float32)
        x0, g0, step = g["init_x0"].astype(np.float32).copy(), g["init_g0"].astype(np.float32).copy(), np.zeros((b, v), np.float32)
        for it in range(iters):
            oracle.lbfgs_step(step, rho, y, s, np.ascontiguousarray(g["q"][it]), np.ascontiguousarray(g["g"][it]), x0, g0, 0.01, True)
            scale = max(float(np.abs(g["step"][it]).max()), 1e-12)
            np.testing.assert_allclose(step, g["step"][it], atol=3e-4 * scale, rtol=3e-3, err_msg=f"step, iteration {it}")
        np.testing.assert_allclose(y, g["y"], atol=1e-6 * max(1.0, float(np.abs(g["y"]).max())), err_msg="y history")
        np.testing.assert_allclose(s, g["s"], atol=1e-6 * max(1.0, float(np.abs(g["s"]).max())), err_msg="s history")
        np.testing.assert_allclose(rho, g["rho"], rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(g["rho"]).max())), err_msg="rho")
    except AssertionError as e:
        bad += 1
        print(f"FAILED L-BFGS case {case}: b {b} v {v} m {m} iterations {iters}: {str(e)[:300]}".replace("\n", " | "))
    kind = str(rng.choice(["wolfe", "strong_wolfe", "approx_wolfe"]))
    b, v = int(rng.integers(1, 300)), int(rng.choice([1, 7, 12, 84]))
    try:
        g = G.line_search_case(rng, b, v, kind)
        x_set, d, c, g_x, al = (g[k] for k in ("x_set", "d", "c", "g_x", "alphas"))
        nls = x_set.shape[1]
        st = dict(best_cost=np.full((b,), 1e9, np.float32), best_action=np.zeros((b, v), np.float32), best_iteration=np.zeros((b,), np.int16),
                  current_iteration=np.zeros((b,), np.int16), converged=np.zeros((b,), np.uint8), exploration_cost=np.zeros((b,), np.float32),
                  exploration_action=np.zeros((b, v), np.float32), exploration_gradient=np.zeros((b, v), np.float32), cost=np.zeros((b,), np.float32),
                  action=np.zeros((b, v), np.float32), gradient=np.zeros((b, v), np.float32), exploration_idx=np.zeros((b, nls), np.int32),
                  selected_idx=np.zeros((b, nls), np.int32))
        oracle.line_search(st, c, x_set, g_x, d, al, 1e-5, 0.9, kind == "strong_wolfe", kind == "approx_wolfe", 5, 0.0, 0.001)
        assert np.array_equal(st["exploration_idx"][:, 0], g["exploration"]), "exploration index"
        np.testing.assert_array_equal(st["exploration_cost"], g["exploration_cost"])
        np.testing.assert_array_equal(st["exploration_action"], g["exploration_action"])
        np.testing.assert_array_equal(st["exploration_gradient"], g["exploration_gradient"])
        # (the CUDA kernel falls back to the Armijo-only index where the twin keeps 0: line_search_helpers.cuh:46-60 vs
        # line_search_strategy.py:622-631; wherever the twin found a full-Wolfe step they agree; strong Wolfe: everywhere)
        sel, tsel = st["selected_idx"][:, 0], g["torch_selected"]
        assert np.array_equal(sel[tsel > 0], tsel[tsel > 0]), "selected index where the twin found a step"
        if kind == "strong_wolfe":
            assert np.array_equal(sel, tsel), "selected index (strong Wolfe)"
    except AssertionError as e:
        bad += 1
        print(f"FAILED line search case {case}: {kind} b {b} v {v}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
