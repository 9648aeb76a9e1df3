"""Optimiser-side kernels against the oracle on random and degenerate inputs: the L-BFGS step (zero / tiny / huge curvature
pairs, empty history, first iterations) and the Wolfe line search (ties between candidates, flat and rising costs): integer
outputs exact.   python tests/randomised/fuzz_opt.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from curobo_amd.backends import optimization as Op  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
# ---------------------------------------------------------------- L-BFGS step
for case in range(n_cases):
    b, v, m = int(rng.integers(1, 200)), int(rng.choice([1, 7, 14, 84, 96, 301])), int(rng.choice([0, 1, 2, 5, 15, 27, 31]))
    mk = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
    mode = int(rng.integers(0, 6))
    st = dict(step=np.zeros((b, v), np.float32), rho=mk(m, b) * 0.1, y=mk(m, b, v), s=mk(m, b, v), q=mk(b, v), g=mk(b, v), x0=mk(b, v), g0=mk(b, v))
    if mode == 1 and m:  # an empty history
        st["rho"][:] = 0; st["y"][:] = 0; st["s"][:] = 0
    if mode == 2:  # no movement since the last iteration on some problems: y = s = 0 for the new pair
        rows = rng.random(b) < 0.5
        st["x0"][rows] = st["q"][rows]; st["g0"][rows] = st["g"][rows]
    if mode == 3:  # tiny / huge scales
        sc = np.float32(10.0 ** rng.integers(-12, 12))
        for k in ("y", "g", "g0"):
            st[k] *= sc
    if mode == 4:  # negative curvature on the new pair
        st["g0"] = st["g"] + (st["q"] - st["x0"])
    if mode == 5:  # zero gradient
        st["g"][rng.random(b) < 0.5] = 0
    dv = {k: torch.as_tensor(a.copy(), device=dev) for k, a in st.items()}
    stable = bool(rng.random() < 0.5)
    try:
        for it in range(3):
            oracle.lbfgs_step(st["step"], st["rho"], st["y"], st["s"], st["q"], st["g"], st["x0"], st["g0"], 0.01, stable)
            Op.launch_lbfgs_step(dv["step"], dv["rho"], dv["y"], dv["s"], dv["q"], dv["g"], dv["x0"], dv["g0"], 0.01, b, m, v, stable, True)
            torch.cuda.synchronize()
            for k in ("step", "rho", "y", "s", "x0", "g0"):
                got, want = dv[k].cpu().numpy(), st[k]
                fin = np.isfinite(want)
                assert np.array_equal(np.isfinite(got), fin), f"{k}: finite pattern differs (it {it})"
                scale = max(1.0, float(np.abs(want[fin]).max())) if fin.any() else 1.0
                np.testing.assert_allclose(got[fin], want[fin], atol=3e-5 * scale, rtol=3e-4, err_msg=f"{k} it{it}")
            st["q"] = (st["q"] + 0.1 * np.nan_to_num(st["step"]) / max(1.0, float(np.abs(np.nan_to_num(st["step"])).max()))).astype(np.float32)
            st["g"] = mk(b, v)
            dv["q"], dv["g"] = torch.as_tensor(st["q"], device=dev), torch.as_tensor(st["g"], device=dev)
    except AssertionError as e:
        bad += 1
        print(f"L-BFGS step FAILED: b {b} v {v} m {m} mode {mode} stable {stable}: {str(e)[:400]}".replace("\n", " | "))
print("L-BFGS step cases:", n_cases, "failed so far:", bad)
# ---------------------------------------------------------------- line search
for case in range(n_cases):
    b, nls, v = int(rng.integers(1, 400)), 4, int(rng.choice([7, 84, 96]))
    x = rng.normal(size=(b, 1, v)).astype(np.float32)
    d = rng.normal(size=(b, 1, v)).astype(np.float32)
    alphas = np.array([0.0, 0.1, 0.5, 1.0], np.float32)
    sa = (x + alphas[None, :, None] * d).astype(np.float32)
    curv = rng.uniform(0.1, 3.0, size=(b, 1, v)).astype(np.float32)
    sg = (curv * sa + 0.3 * rng.normal(size=sa.shape)).astype(np.float32)
    scost = (0.5 * (curv * sa * sa).sum(-1, keepdims=True) + rng.normal(size=(b, nls, 1))).astype(np.float32)
    mode = int(rng.integers(0, 5))
    if mode == 1:  # all candidates of a problem cost the same
        rows = rng.random(b) < 0.5
        scost[rows] = scost[rows][:, :1]
    if mode == 2:  # two candidates tie for the minimum
        scost[:, 2] = scost[:, 1]
    if mode == 3:  # the cost rises along the direction (no candidate satisfies Armijo)
        scost = np.sort(scost, axis=1)
    if mode == 4:  # zero direction
        d[rng.random(b) < 0.5] = 0
        sa = (x + alphas[None, :, None] * d).astype(np.float32)
    strong, approx = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    if strong:
        approx = False
    st = dict(best_cost=(np.full((b,), 1e3, np.float32) * rng.uniform(0, 1, size=b).astype(np.float32)), best_action=np.zeros((b, v), np.float32),
              best_iteration=np.zeros((b,), np.int16), current_iteration=rng.integers(0, 30, size=b).astype(np.int16),
              converged=np.zeros((b,), np.uint8), exploration_cost=np.zeros((b,), np.float32), exploration_action=np.zeros((b, v), np.float32),
              exploration_gradient=np.zeros((b, v), np.float32), cost=np.zeros((b,), np.float32), action=np.zeros((b, v), np.float32),
              gradient=np.zeros((b, v), np.float32), exploration_idx=np.zeros((b, nls), np.int32), selected_idx=np.zeros((b, nls), np.int32))
    dv = {k: torch.as_tensor(a.copy(), device=dev) for k, a in st.items()}
    oracle.line_search(st, scost, sa, sg, d, alphas, 1e-5, 0.9, strong, approx, 5, 1e-4, 1e-3)
    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    Op.launch_line_search(dv["best_cost"], dv["best_action"], dv["best_iteration"], dv["current_iteration"], dv["converged"], 5, 1e-4, 1e-3,
                          dv["exploration_cost"], dv["exploration_action"], dv["exploration_gradient"], dv["exploration_idx"], dv["cost"],
                          dv["action"], dv["gradient"], dv["selected_idx"], t(scost), t(sa), t(sg), t(d), t(alphas), 1e-5, 0.9, strong, approx,
                          nls, v, b)
    torch.cuda.synchronize()
    try:
        for k in ("selected_idx", "exploration_idx", "best_iteration", "current_iteration", "converged"):
            assert np.array_equal(dv[k].cpu().numpy(), st[k]), f"{k}: {int((dv[k].cpu().numpy() != st[k]).sum())} integer outputs differ"
        for k in ("best_cost", "best_action", "exploration_cost", "exploration_action", "exploration_gradient", "cost", "action", "gradient"):
            np.testing.assert_array_equal(dv[k].cpu().numpy(), st[k], err_msg=k)
    except AssertionError as e:
        bad += 1
        print(f"line search FAILED: b {b} v {v} mode {mode} strong {strong} approx {approx}: {str(e)[:400]}".replace("\n", " | "))
print("line search cases:", n_cases, "failed in total:", bad)
