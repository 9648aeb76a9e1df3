"""The optimiser stage on the CPU with torch: history update + two-loop recursion over a batch of problems.

TEST / BASELINE INFRASTRUCTURE (see oracle/__init__.py): a torch restatement, in this repository's own words, of what the
reference's torch fallbacks compute (curobo/_src/optim/gradient/lbfgs_jit_helpers.py:10-78: ``jit_lbfgs_update_buffers``
then ``jit_lbfgs_compute_step_direction``; ``LBFGSOpt._get_step_direction``, optim/gradient/lbfgs.py:156-204 calls them in
that order).  SURVEY.md section 8(d)(2) asks for the optimiser-stage CPU baseline to be those torch functions on
``device="cpu"``; the reference tree does not travel to the GPU box, so this twin is timed there instead
(``bench.py`` ``cpu_baseline.optimizer_stage.torch_twin``).  Pinned by the golden the reference's own functions produced
here (tests/golden/optim_golden.npz, tests/test_oracle_optim.py).  Layout: history [m, B, V], rho [m, B], x / g [B, V].
"""

import torch


def lbfgs_step(rho, y, s, x, g, x0, g0, epsilon: float = 0.01, stable_mode: bool = True):
    """in place: append (s, y, rho) of the step x0 -> x to the history (oldest slot first, newest last), x0 / g0 <- x / g;
    returns the new search direction [B, V]"""
    yk, sk = g - g0, x - x0
    r = 1.0 / (yk * sk).sum(-1)
    if stable_mode:
        r = torch.nan_to_num(r, 0.0, 0.0, 0.0)
    y.copy_(torch.roll(y, -1, 0)); y[-1] = yk
    s.copy_(torch.roll(s, -1, 0)); s[-1] = sk
    rho.copy_(torch.roll(rho, -1, 0)); rho[-1] = r
    x0.copy_(x); g0.copy_(g)
    m = y.shape[0]
    q = g.clone()
    alpha = torch.zeros_like(rho)
    for i in range(m - 1, -1, -1):
        alpha[i] = rho[i] * (s[i] * q).sum(-1)
        q = q - alpha[i].unsqueeze(-1) * y[i]
    gamma = (s[-1] * y[-1]).sum(-1) / (y[-1] * y[-1]).sum(-1)
    if stable_mode:
        gamma = torch.nan_to_num(gamma, epsilon, epsilon, epsilon)
    z = torch.relu(gamma).unsqueeze(-1) * q
    for i in range(m):
        beta = rho[i] * (y[i] * z).sum(-1)
        z = z + (alpha[i] - beta).unsqueeze(-1) * s[i]
    return -z
