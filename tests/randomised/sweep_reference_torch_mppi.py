"""The NumPy restatement of the MPPI distribution update (oracle/mppi_ref.py) against the reference's own torch functions
(curobo/_src/optim/particle/mppi.py: jit_mean_cov_diag_a, jit_calculate_exp_util_from_costs) on random shapes and
temperatures.  CPU only.   python tests/randomised/sweep_reference_torch_mppi.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/optim/particle"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from oracle.mppi_ref import mean_cov_diag_a  # noqa: E402

sys.path.insert(0, "/root/reference")
from curobo._src.optim.particle import mppi as ref  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    b, p, h, ha, d = int(rng.integers(1, 6)), int(rng.choice([1, 2, 37, 64, 500])), int(rng.choice([1, 16, 33])), int(rng.choice([1, 12, 30])), int(rng.integers(1, 10))
    beta, gamma = float(rng.choice([0.02, 0.1, 1.0, 10.0])), float(rng.choice([1.0, 0.97, 0.8]))
    sm, sc, kappa = float(rng.choice([0.0, 0.5, 0.9, 1.0])), float(rng.choice([0.0, 0.1, 1.0])), float(rng.choice([0.0, 0.01]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    costs = torch.rand(b, p, h, generator=g) * float(rng.choice([0.1, 5.0, 100.0]))
    actions = torch.randn(b, p, ha, d, generator=g)
    mean = torch.randn(b, ha, d, generator=g) * 0.3
    cov = torch.rand(b, 1, d, generator=g) + 0.1
    gamma_seq = torch.cumprod(torch.full((1, 1, h), gamma), dim=-1)
    try:
        new_mean, new_cov, new_tril = ref.jit_mean_cov_diag_a(costs, actions, gamma_seq, mean, cov, sm, sc, kappa, beta)
        w = ref.jit_calculate_exp_util_from_costs(costs, gamma_seq, beta)
        m2, c2, t2, w2, _ = mean_cov_diag_a(costs.numpy(), actions.numpy(), gamma_seq.numpy(), mean.numpy(), cov.numpy(), sm, sc, kappa, beta)
        # a sharp softmax (total cost / beta in the hundreds or thousands) multiplies the rounding of the cost sums: exp(x (1 + eps))
        sharp = float((costs * gamma_seq).sum(-1).max()) / beta
        k = max(1.0, sharp * 2e-3)
        np.testing.assert_allclose(w2, w.numpy(), rtol=5e-4 * k, atol=2e-7, err_msg="weights")
        np.testing.assert_allclose(m2, new_mean.numpy(), rtol=2e-5 * k, atol=4e-5 * k, err_msg="mean")
        np.testing.assert_allclose(c2, new_cov.numpy(), rtol=2e-5 * k, atol=4e-5 * k, err_msg="cov")
        np.testing.assert_allclose(t2, new_tril.numpy(), rtol=2e-5 * k, atol=4e-5 * k, err_msg="tril")
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: b {b} p {p} h {h} ha {ha} d {d} beta {beta} gamma {gamma} sm {sm} sc {sc} kappa {kappa}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
