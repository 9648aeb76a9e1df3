"""The HIP MPPI distribution update against the NumPy restatement of the reference's torch functions (oracle/mppi_ref.py) on
random shapes and temperatures: weights, new mean / covariance / scale, the best particle (exact).
    python tests/randomised/fuzz_mppi.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from curobo_amd.backends import optimization as Op  # noqa: E402
from oracle.mppi_ref import mean_cov_diag_a  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
bad = 0
for case in range(n_cases):
    b, p, h, ha, d = int(rng.integers(1, 20)), int(rng.choice([1, 2, 37, 64, 500, 1024])), int(rng.choice([1, 16, 33, 65])), int(rng.choice([1, 12, 30])), int(rng.integers(1, 50))
    beta, gamma = float(rng.choice([0.02, 0.1, 1.0, 10.0])), float(rng.choice([1.0, 0.97, 0.8]))
    sm, sc, kappa = float(rng.choice([0.0, 0.5, 0.9, 1.0])), float(rng.choice([0.0, 0.1, 1.0])), float(rng.choice([0.0, 0.01]))
    costs = (rng.random((b, p, h)) * float(rng.choice([0.1, 5.0, 100.0]))).astype(np.float32)
    actions = rng.standard_normal((b, p, ha, d)).astype(np.float32)
    mean = (rng.standard_normal((b, ha, d)) * 0.3).astype(np.float32)
    cov = (rng.random((b, 1, d)) + 0.1).astype(np.float32)
    gamma_seq = np.cumprod(np.full((1, 1, h), gamma, np.float32), axis=-1)
    try:
        m2, c2, t2, w2, best2 = mean_cov_diag_a(costs, actions, gamma_seq, mean, cov, sm, sc, kappa, beta)
        new_mean, new_cov, new_tril = torch.zeros(b, ha, d, device=dev), torch.zeros(b, 1, d, device=dev), torch.zeros(b, 1, d, device=dev)
        best, w = torch.zeros(b, ha, d, device=dev), torch.zeros(b, p, device=dev)
        Op.mppi_update_distribution(new_mean, new_cov, new_tril, best, w, t(costs), t(gamma_seq.reshape(-1)), t(actions), t(mean), t(cov), beta, sm, sc, kappa)
        torch.cuda.synchronize()
        # (a weight is exp(-(total - min) / beta): the rounding of an fp32 total, ~2^-24 of its size, is divided by beta)
        w_rtol = 1e-3 + 2.0 * 6e-8 * float(np.abs((costs * gamma_seq).sum(-1)).max()) / beta
        np.testing.assert_allclose(w.cpu().numpy(), w2, rtol=w_rtol, atol=2e-7, err_msg="weights")
        np.testing.assert_allclose(new_mean.cpu().numpy(), m2, rtol=2e-4, atol=5e-5, err_msg="mean")
        np.testing.assert_allclose(new_cov.cpu().numpy(), c2, rtol=2e-4, atol=5e-5, err_msg="cov")
        np.testing.assert_allclose(new_tril.cpu().numpy(), t2, rtol=2e-4, atol=5e-5, err_msg="tril")
        # the best particle: exact unless two particles tie in total cost to rounding
        tot = (costs * gamma_seq).sum(-1)
        srt = np.sort(tot, axis=1)
        clear = (srt[:, 1] - srt[:, 0] > 1e-5 * np.abs(srt[:, 0])) if p > 1 else np.ones(b, bool)
        assert np.array_equal(best.cpu().numpy()[clear], best2[clear]), "best particle"
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: b {b} p {p} h {h} ha {ha} d {d} beta {beta} gamma {gamma} sm {sm} sc {sc} kappa {kappa}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
