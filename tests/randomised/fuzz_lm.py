"""The HIP Levenberg-Marquardt step (J^T J on the matrix cores, Cholesky in LDS) against a float64 solve and the oracle on random
systems: 1-64 dof, 1-80 residuals, batch 1-300, damping 1e-4 .. 1e2, well and badly scaled Jacobians.
    python tests/randomised/fuzz_lm.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from curobo_amd.backends import linalg as La  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
bad = 0
for case in range(n_cases):
    dof = int(rng.choice([1, 2, 3, 6, 7, 12, 15, 16, 17, 31, 32, 33, 49, 64]))
    n_res = int(rng.choice([1, 3, 6, 13, 16, 17, 20, 33, 64, 80]))
    b = int(rng.choice([1, 2, 37, 64, 255, 300]))
    J = (rng.normal(size=(b, n_res, dof)) * 10.0 ** rng.uniform(-1, 1)).astype(np.float32)
    g = np.einsum("brd,br->bd", J, rng.normal(size=(b, n_res))).astype(np.float32)
    # damping relative to the system's scale, so that the fp32 solve is well posed whatever n_res / dof
    lam = (10.0 ** rng.uniform(-3, 1, size=b) * np.maximum(np.einsum("brd,brd->b", J, J) / dof, 1e-6)).astype(np.float32)
    q = rng.normal(size=(b, dof)).astype(np.float32)
    try:
        q_out, pred = torch.zeros(b, dof, device=dev), torch.zeros(b, device=dev)
        La.levenberg_marquardt_step(q_out, pred, t(J), t(g), t(lam), t(q))
        torch.cuda.synchronize()
        J64, g64 = J.astype(np.float64), g.astype(np.float64)
        A = np.einsum("brd,bre->bde", J64, J64) + lam[:, None, None].astype(np.float64) * np.eye(dof)
        delta = np.linalg.solve(A, -g64[..., None])[..., 0]
        got = q_out.cpu().numpy() - q
        np.testing.assert_allclose(got, delta, rtol=5e-3, atol=2e-3 * max(1e-12, float(np.abs(delta).max())), err_msg="delta vs the float64 solve")
        pred64 = 0.5 * np.einsum("bd,bd->b", delta, lam[:, None] * delta - g64)
        np.testing.assert_allclose(pred.cpu().numpy(), pred64, rtol=1e-2, atol=2e-3 * max(1e-12, float(np.abs(pred64).max())), err_msg="pred_reduction")
        q_ref, _ = oracle.lm_step(J, g, lam, q)
        np.testing.assert_allclose(got, q_ref - q, rtol=5e-3, atol=2e-3 * max(1e-12, float(np.abs(q_ref - q).max())), err_msg="delta vs the oracle")
    except (AssertionError, ValueError) as ex:
        bad += 1
        print(f"FAILED case {case}: dof {dof} residuals {n_res} batch {b}: {type(ex).__name__} {str(ex)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
