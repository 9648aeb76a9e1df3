"""dJ/dq (the gradient THROUGH the geometric-Jacobian output) against the reference's own CUDA kernel.

``tests/golden/jacobian_grad_golden.npz`` was produced by ``kinematics_backward_kernel<..., COMPUTE_JACOBIAN_GRAD = true>`` run on
the CPU (``oracle/_ref``, generator ``tests/golden/make_jacobian_grad_golden.py``).  CPU: the golden is consistent with finite
differences of the reference Jacobian.  GPU: the HIP kernel reproduces it (until now it was checked against finite differences
only, at 2e-2): 1e-4 of the largest entry, Jacobian itself 1e-5.
"""
import os

import numpy as np
import pytest

from conftest import load_model

from oracle import ref_kernels

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobian_grad_golden.npz")


@pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref/libcurobo_ref.so not built")
@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_golden_is_the_derivative_of_the_reference_jacobian(robot):
    g = np.load(GOLD)
    ref = ref_kernels.ReferenceKernels()
    md = load_model(robot).as_dict()
    q = g[f"{robot}/q"].reshape(6, -1)
    w = g[f"{robot}/w"].reshape(6, *g[f"{robot}/w"].shape[2:]).astype(np.float64)
    got = g[f"{robot}/grad_q"].reshape(6, -1).astype(np.float64)
    eps, fd = 1e-3, np.zeros_like(got)
    for d in range(q.shape[1]):
        dq = np.zeros_like(q)
        dq[:, d] = eps
        jp = ref.kinematics_forward(q + dq, md, compute_jacobian=True)["jacobian"].astype(np.float64)
        jm = ref.kinematics_forward(q - dq, md, compute_jacobian=True)["jacobian"].astype(np.float64)
        fd[:, d] = (((jp - jm) / (2 * eps)) * w).sum(axis=(1, 2, 3))
    np.testing.assert_allclose(got, fd, rtol=0, atol=2e-3 * np.abs(fd).max())


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_hip_reproduces_the_reference_jacobian_gradient(robot, device):
    import torch

    from curobo_amd.kinematics import Kinematics, KinematicsCfg

    g = np.load(GOLD)
    cfg = KinematicsCfg.from_packaged(robot, device=device)
    kin = Kinematics(cfg, compute_jacobian=True, compute_spheres=False)
    q = torch.as_tensor(g[f"{robot}/q"], device=device).requires_grad_(True)
    w = torch.as_tensor(g[f"{robot}/w"], device=device)
    st = kin.compute_kinematics(q)
    np.testing.assert_allclose(st.tool_jacobians.detach().cpu().numpy(), g[f"{robot}/jacobian"], rtol=0, atol=1e-5)
    (st.tool_jacobians * w).sum().backward()
    want = g[f"{robot}/grad_q"]
    np.testing.assert_allclose(q.grad.cpu().numpy(), want, rtol=0, atol=1e-4 * np.abs(want).max())
