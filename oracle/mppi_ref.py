"""NumPy restatement of the MPPI distribution update (test infrastructure, like the rest of
``oracle/``): reference ``curobo/_src/optim/particle/mppi.py`` ``jit_calculate_exp_util_from_costs``
(:638-656), ``jit_blend_mean`` (:708-724), ``jit_diag_a_cov_update`` (:666-684), ``jit_blend_cov``
(:687-705), ``jit_mean_cov_diag_a`` (:727-757) and the BEST sample of ``_update_distribution``
(:217-220).  Pinned by ``tests/golden/mppi_golden.npz`` (the reference's own torch functions run on
CPU, ``tests/golden/make_mppi_golden.py``)."""

import numpy as np


def exp_util_from_costs(costs, gamma_seq, beta):
    """costs [b, p, h], gamma_seq [1, 1, h] -> softmax weights [b, p]"""
    total = (gamma_seq * costs).sum(-1) / gamma_seq[..., 0]
    x = (-1.0 / beta) * total
    x = x - x.max(-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(-1, keepdims=True)


def mean_cov_diag_a(costs, actions, gamma_seq, mean, cov, step_size_mean, step_size_cov, kappa, beta):
    """actions [b, p, ha, d], mean [b, ha, d], cov [b, 1, d] -> (new_mean, new_cov, new_scale_tril, w, best)"""
    f = np.float32
    costs, actions, gamma_seq, mean, cov = (np.asarray(x, f) for x in (costs, actions, gamma_seq, mean, cov))
    w = exp_util_from_costs(costs, gamma_seq, f(beta)).astype(f)
    w4 = w[..., None, None]
    new_mean = (f(1.0) - f(step_size_mean)) * mean + f(step_size_mean) * (w4 * actions).sum(-3)
    delta = actions - mean[:, None]
    cov_update = (w4 * delta ** 2).sum(-3).mean(-2)[:, None]
    new_cov = (f(1.0) - f(step_size_cov)) * cov + f(step_size_cov) * cov_update + f(kappa)
    best = actions[np.arange(actions.shape[0]), np.argmax(w, -1)]
    return new_mean.astype(f), new_cov.astype(f), np.sqrt(new_cov).astype(f), w, best
