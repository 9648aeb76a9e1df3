"""The NumPy MPPI restatement against golden vectors from the reference's own torch functions
(tests/golden/make_mppi_golden.py runs curobo/_src/optim/particle/mppi.py on CPU)."""

import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mppi_golden.npz")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_mppi_update_matches_reference_torch(case):
    from oracle.mppi_ref import mean_cov_diag_a

    g = np.load(GOLD)
    k = lambda n: g[f"c{case}/{n}"]  # noqa: E731
    sm, sc, kappa, beta = [float(x) for x in k("params")]
    new_mean, new_cov, new_tril, w, _ = mean_cov_diag_a(k("costs"), k("actions"), k("gamma_seq"), k("mean"), k("cov"),
                                                        sm, sc, kappa, beta)
    np.testing.assert_allclose(w, k("w"), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(new_mean, k("new_mean"), rtol=1e-5, atol=2e-5)  # sharp softmax (beta 0.05): fp32 exp
    np.testing.assert_allclose(new_cov, k("new_cov"), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(new_tril, k("new_tril"), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/curobo/_src/optim/particle"), reason="the reference's torch functions are not on this machine")
def test_randomised_sweep_against_the_reference_torch_functions():
    """tests/randomised/sweep_reference_torch_mppi.py at a small size: random shapes, temperatures, step sizes"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", "sweep_reference_torch_mppi.py"), "60", "9"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]
