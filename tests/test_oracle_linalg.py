"""The Levenberg-Marquardt step oracle against numpy.linalg.solve in float64 (the reference's Warp
tile kernel cannot run here: SURVEY 8c marks this parity as pinned against numpy)."""

import numpy as np
import pytest


@pytest.mark.parametrize("dof,n_res", [(7, 13), (6, 12), (49, 73), (16, 20), (33, 40)])
def test_lm_step_solves_damped_normal_equations(dof, n_res, oracle):
    rng = np.random.default_rng(dof)
    b = 9
    J = rng.normal(size=(b, n_res, dof)).astype(np.float32)
    r = rng.normal(size=(b, n_res)).astype(np.float32)
    g = np.einsum("brd,br->bd", J, r).astype(np.float32)
    lam = rng.uniform(1e-3, 1.0, size=b).astype(np.float32)
    q = rng.normal(size=(b, dof)).astype(np.float32)
    q_out, pred = oracle.lm_step(J, g, lam, q)
    J64, g64 = J.astype(np.float64), g.astype(np.float64)
    A = np.einsum("brd,bre->bde", J64, J64) + lam[:, None, None].astype(np.float64) * np.eye(dof)
    delta = np.linalg.solve(A, -g64[..., None])[..., 0]
    np.testing.assert_allclose(q_out - q, delta, rtol=2e-3, atol=2e-4 * np.abs(delta).max())
    np.testing.assert_allclose(pred, 0.5 * (delta * (lam[:, None] * delta - g64)).sum(1), rtol=2e-3, atol=1e-4)
    # -g = A delta with A positive definite, so delta.(lambda delta - g) = lambda |delta|^2 + delta^T A delta >= 0
    assert np.all(pred >= -1e-6)
