"""HIP backend for the dense linear algebra of the seed-IK solver.

The reference runs the Levenberg-Marquardt step as an NVIDIA Warp tile kernel without a backend
hook (``curobo/_src/optim/util/levenberg_marquardt_step.py:96-199``); this is the replacement body of
``LevenbergMarquardtStep.__call__`` (same tensors, mutated in place, current stream)."""

from __future__ import annotations

import torch

from .._lib import check, current_stream, load, ptr


def levenberg_marquardt_step(
    joint_position_out: torch.Tensor,
    pred_reduction: torch.Tensor,
    jacobian: torch.Tensor,
    jTerror: torch.Tensor,
    lambda_damping: torch.Tensor,
    joint_position_in: torch.Tensor,
):
    """jacobian [b, n_residuals, action_dim]; J^T J on the matrix cores, Cholesky solve in LDS."""
    b, r, d = jacobian.shape
    check(load().curobo_hip_levenberg_marquardt_step(
        ptr(joint_position_out), ptr(pred_reduction), ptr(jacobian), ptr(jTerror), ptr(lambda_damping),
        ptr(joint_position_in), b, r, d, current_stream(joint_position_out),
    ))
