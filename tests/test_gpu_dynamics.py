"""RNEA HIP kernels vs the oracle (itself pinned by the reference's NumPy implementation)."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q

pytestmark = pytest.mark.gpu


def _setup(robot, device, n, seed):
    from curobo_amd.robot.kinematics_params import KinematicsParams

    model = load_model(robot)
    kin = KinematicsParams.from_model(model, device)
    rng = np.random.default_rng(seed)
    q = sample_q(model, n, seed=seed).astype(np.float32)
    qd = rng.normal(size=q.shape).astype(np.float32)
    qdd = (rng.normal(size=q.shape) * 2).astype(np.float32)
    return model, kin, q, qd, qdd, rng


@pytest.mark.parametrize("robot,n", [("franka", 1000), ("ur10e", 257), ("unitree_g1", 300)])
@pytest.mark.parametrize("with_f_ext", [False, True])
def test_rnea_forward_backward_match_oracle(robot, n, with_f_ext, oracle, device):
    from curobo_amd.backends import dynamics as Dy

    model, kin, q, qd, qdd, rng = _setup(robot, device, n, 3)
    md = model.as_dict()
    L, D = kin.num_links, kin.num_dof
    fe = rng.normal(size=(n, L, 6)).astype(np.float32) if with_f_ext else None
    grav = np.array([0, 0, 0, 0, 0, 9.81], np.float32)
    tau_ref, cache_ref = oracle.rnea_forward(q, qd, qdd, md, gravity=grav, f_ext=fe)
    t = lambda a: None if a is None else torch.as_tensor(a, device=device)  # noqa: E731
    tau = torch.zeros(n, D, device=device)
    cache = torch.zeros(n, L * 20, device=device)
    args = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
            kin.joint_offset_map, t(grav), kin.link_level_offsets, kin.link_level_data)
    Dy.launch_rnea_forward(tau, t(q), t(qd), t(qdd), *args, cache, n, L, D, kin.n_tree_levels, 1, t(fe))
    torch.cuda.synchronize()
    scale = max(1.0, np.abs(tau_ref).max())
    np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, rtol=1e-4, atol=1e-5 * scale)
    # the cache is opaque to callers; its internal layout is [link][20][batch]
    got = cache.cpu().numpy().reshape(L, 20, n).transpose(2, 0, 1)
    for sl in (slice(0, 6), slice(6, 12), slice(12, 18)):
        ref = cache_ref[:, :, sl]
        np.testing.assert_allclose(got[:, :, sl], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
    w = rng.normal(size=(n, D)).astype(np.float32)
    ref_g = oracle.rnea_backward(w, q, qd, cache_ref, md, gravity=grav, want_f_ext_grad=with_f_ext)
    g = [torch.full((n, D), 7.0, device=device) for _ in range(3)]  # must be fully rewritten
    gfe = torch.zeros(n, L, 6, device=device) if with_f_ext else None
    Dy.launch_rnea_backward(*g, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, gfe)
    torch.cuda.synchronize()
    for ours, ref in zip(g, ref_g[:3]):
        np.testing.assert_allclose(ours.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(ref).max()))
    if with_f_ext:
        np.testing.assert_allclose(gfe.cpu().numpy(), ref_g[3], rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref_g[3]).max()))


def test_dynamics_front_end_autograd(oracle, device):
    """Dynamics.compute_inverse_dynamics: shapes, autograd through the HIP VJP, energy-style cost"""
    from curobo_amd.dynamics import Dynamics

    model, kin, q, qd, qdd, rng = _setup("franka", device, 12 * 5, 5)
    dyn = Dynamics(kin)
    tq = [torch.as_tensor(x, device=device).reshape(12, 5, 7).requires_grad_(True) for x in (q, qd, qdd)]
    tau = dyn.compute_inverse_dynamics(*tq)
    assert tau.shape == (12, 5, 7)
    tau_ref, cache_ref = oracle.rnea_forward(q, qd, qdd, model.as_dict())
    np.testing.assert_allclose(tau.detach().cpu().numpy().reshape(-1, 7), tau_ref, rtol=1e-4, atol=1e-4)
    (tau ** 2).sum().backward()
    ref = oracle.rnea_backward(2 * tau_ref, q, qd, cache_ref, model.as_dict())
    for x, r in zip(tq, ref):
        np.testing.assert_allclose(x.grad.cpu().numpy().reshape(-1, 7), r, rtol=2e-3, atol=2e-4 * np.abs(r).max())
    # gravity compensation at rest: only gravity torques, base joint (vertical axis) carries none
    z = torch.zeros(4, 7, device=device)
    tg = dyn.compute_inverse_dynamics(torch.as_tensor(q[:4], device=device), z, z)
    assert float(tg[:, 0].abs().max()) < 1e-4 and float(tg.abs().max()) > 1.0
