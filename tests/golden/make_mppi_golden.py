"""Golden vectors for the MPPI distribution update from the REFERENCE's own torch functions
(curobo/_src/optim/particle/mppi.py: jit_mean_cov_diag_a, jit_calculate_exp_util_from_costs), run on
CPU in the build container.   PYTHONPATH=/root/reference python tests/golden/make_mppi_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from curobo._src.optim.particle import mppi as ref

    out = {}
    for ci, (b, p, h, ha, d, beta, gamma) in enumerate([(3, 64, 33, 12, 7, 0.1, 1.0), (2, 500, 1, 30, 6, 1.0, 1.0),
                                                         (4, 37, 16, 16, 9, 0.05, 0.97)]):
        g = torch.Generator().manual_seed(ci)
        costs = torch.rand(b, p, h, generator=g) * 5.0
        actions = torch.randn(b, p, ha, d, generator=g)
        mean = torch.randn(b, ha, d, generator=g) * 0.3
        cov = torch.rand(b, 1, d, generator=g) + 0.1
        gamma_seq = torch.cumprod(torch.full((1, 1, h), gamma), dim=-1)
        sm, sc, kappa = 0.9, 0.1, 0.01
        new_mean, new_cov, new_tril = ref.jit_mean_cov_diag_a(costs, actions, gamma_seq, mean, cov, sm, sc, kappa, beta)
        w = ref.jit_calculate_exp_util_from_costs(costs, gamma_seq, beta)
        for k, v in dict(costs=costs, actions=actions, mean=mean, cov=cov, gamma_seq=gamma_seq, new_mean=new_mean,
                         new_cov=new_cov, new_tril=new_tril, w=w, params=torch.tensor([sm, sc, kappa, beta])).items():
            out[f"c{ci}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "mppi_golden.npz"), **out)
    print("wrote mppi_golden.npz", len(out))


if __name__ == "__main__":
    main()
