"""RNEA HIP kernels vs the oracle (itself pinned by the reference's NumPy implementation)."""

import os

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


def _max_err(got, ref, scale=None):
    """largest |got - ref| in units of ``scale`` (default: the largest |ref|) -- the figure the C4 tolerances are set from"""
    ref = np.asarray(ref, np.float64)
    s = float(np.abs(ref).max()) if scale is None else float(scale)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(s, 1e-30))


@pytest.mark.parametrize("scratch", [False, True], ids=["staged", "scratch"])
@pytest.mark.parametrize("B,H", [(4, 6), (256, 33)])
def test_c4_shape_humanoid_self_collision_plus_inverse_dynamics_cost(B, H, scratch, oracle, device):
    """BASELINE config 4, in small and at the size of one GPU's share of the benchmark (256 seeds x 33 points): Unitree G1
    whole body (the in-tree stand-in for the 38-DoF humanoid), map-reduce-sized self collision (162 k sphere pairs, dense
    bitmap + broad-phase tiles) + an inverse-dynamics cost (joint-torque limits and torque regularisation on RNEA's tau,
    through the c-space STATE cost) and the complete VJP chain back to (q, qd, qdd) -- HIP kernels vs the oracle
    composition of the same stages.  ``scratch``: the RNEA launches in their scratch / lane form
    (``rnea_transpose_kernel`` + ``rnea_scratch_kernel``: what ``bench.py`` C4 and ``TrajOptRollout`` run next to the
    collision kernels) instead of the staged quad walks.

    Tolerances (north_star: costs within 1e-5 fp32): torques 1e-5 of the largest torque, per-trajectory cost 1e-5
    relative, gradients 5e-4 of the largest gradient entry.  The measured errors are printed (``pytest -s``) and sit at
    a few 1e-7 / 1e-6 / 1e-5: the bounds are the contract, ~10x what fp32 accumulation over a 56-link tree leaves."""
    from curobo_amd.backends import cost as Cs
    from curobo_amd.backends import dynamics as Dy
    from curobo_amd.backends import geometry as G
    from curobo_amd.backends import kinematics as K

    n = B * H
    model, kin, q, qd, qdd, rng = _setup("unitree_g1", device, n, 11)
    md = model.as_dict()
    L, D, S, T = kin.num_links, kin.num_dof, kin.num_spheres, kin.num_pose_links
    P = model.collision_pairs.shape[0]
    assert P > 100000 and D >= 29
    qd, qdd = 0.5 * qd, 0.5 * qdd
    grav = np.array([0, 0, 0, 0, 0, 9.81], np.float32)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dt)  # noqa: E731
    z = lambda *s: torch.zeros(*s, device=device)  # noqa: E731
    w_self = 3.0
    # ---------------- oracle
    fk = oracle.kinematics_forward(q, md)
    sc = oracle.self_collision(fk["robot_spheres"], model.sphere_padding, model.collision_pairs, w_self)
    tau_ref, cache_ref = oracle.rnea_forward(q, qd, qdd, md, gravity=grav)
    # torque limits chosen so that about half of the joint torques violate them (the robot file's
    # limits are placeholders for the virtual base joints)
    cap = np.float32(np.median(np.abs(tau_ref)) + 1e-3)
    eff_b = np.stack([-cap * np.ones(D, np.float32), cap * np.ones(D, np.float32)])
    lim = {"position": np.asarray(md["joint_limits_position"], np.float32), "effort": eff_b.astype(np.float32)}
    weight = np.array([50.0, 1.0, 1.0, 1.0, 7.0], np.float32)          # position, velocity, acceleration, jerk, effort bounds
    eta = np.array([0.05, 0.0, 0.0, 0.0, 0.05], np.float32)
    sql2 = np.array([0.01, 0.02, 0.0, 0.003, 0.0], np.float32)           # vel, acc, jerk, effort regularisation, (dt)
    dt = np.full(B, 0.1, np.float32)
    shp = (B, H, D)
    cs = oracle.cspace_state_cost(q.reshape(shp), qd.reshape(shp), qdd.reshape(shp), np.zeros(shp, np.float32), dt, lim,
                                  weight, eta, sql2, effort=tau_ref.reshape(shp))
    g_q1 = oracle.kinematics_backward(md, fk["cumul_mat"], sc["gradient"])
    g_rnea = oracle.rnea_backward(cs["grad_effort"].reshape(n, D), q, qd, cache_ref, md, gravity=grav)
    ref_cost = sc["distance"].reshape(B, H).sum(-1) + cs["cost"].reshape(B, -1).sum(-1)
    ref_gq = g_q1 + g_rnea[0] + cs["grad_position"].reshape(n, D)
    ref_gqd = g_rnea[1] + cs["grad_velocity"].reshape(n, D)
    ref_gqdd = g_rnea[2] + cs["grad_acceleration"].reshape(n, D)
    assert (sc["distance"] > 0).any() and (np.abs(cs["grad_effort"]) > 0).any()
    # ---------------- HIP
    tq, tqd, tqdd = t(q), t(qd), t(qdd)
    link_pos, link_quat, spheres, com, cumul = z(n, T, 3), z(n, T, 4), z(n, S, 4), z(n, 4), z(n, L, 3, 4)
    env = torch.zeros(n, dtype=torch.int32, device=device)
    K.launch_kinematics_forward_spheres(link_pos, link_quat, spheres, com, cumul, tq, kin.fixed_transforms, kin.link_spheres,
                                        kin.link_masses_com, kin.joint_map_type, kin.joint_map, kin.link_map, kin.tool_frame_map,
                                        kin.link_sphere_idx_map, kin.joint_offset_map, env, kin.num_envs, n, 1, D, S, 32, True, False)
    self_d, self_g, flags = z(n, 1), z(n, S, 4), torch.zeros(n, S, dtype=torch.uint8, device=device)
    scp = kin.self_collision
    G.self_collision_distance(self_d, self_g, z(1), flags, spheres, scp.sphere_padding, torch.tensor([w_self], device=device),
                              scp.collision_pairs, z(1), torch.zeros(2, dtype=torch.int16, device=device), 1, 256, n, 1, S, P,
                              False, True)
    tau, cache = z(n, D), z(n, L * 20)
    rargs = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
             kin.joint_offset_map, t(grav), kin.link_level_offsets, kin.link_level_data)
    rnea_scratch = torch.full((3 * n * D,), float("nan"), device=device) if scratch else None
    Dy.launch_rnea_forward(tau, tq, tqd, tqdd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=rnea_scratch)
    big = np.stack([-1e9 * np.ones(D), 1e9 * np.ones(D)]).astype(np.float32)
    c_cost, gp, gv, ga, gj, gtau = [z(B, H, D) for _ in range(6)]
    Cs.cspace_state_cost(c_cost, gp, gv, ga, gj, gtau, tq.view(shp), tqd.view(shp), tqdd.view(shp), z(*shp), tau.view(shp), t(dt),
                         z(1, D), torch.zeros(B, dtype=torch.int32, device=device), t(lim["position"]), t(big), t(big), t(big),
                         t(lim["effort"]), t(weight), t(eta), t(sql2), z(1), torch.ones(1, device=device), torch.ones(D, device=device),
                         True, B, H, D)
    g1 = z(n, D)
    K.launch_kinematics_backward(g1, z(n, T, 3), z(n, T, 4), self_g, com, com, z(n, T, 3), cumul, kin.link_spheres,
                                 kin.link_masses_com, kin.link_map, kin.joint_map, kin.joint_map_type, kin.tool_frame_map,
                                 kin.link_sphere_idx_map, kin.link_chain_data, kin.link_chain_offsets, kin.joint_links_data,
                                 kin.joint_links_offsets, kin.joint_affects_endeffector, kin.joint_offset_map, env, kin.num_envs,
                                 n, 1, D, S, False, False)
    g2 = [z(n, D) for _ in range(3)]
    # (the scratch VJP as TrajOptRollout launches it: q / qd still in the forward launch's scratch)
    Dy.launch_rnea_backward(*g2, gtau.view(n, D), tq, tqd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None,
                            scratch=rnea_scratch, scratch_holds_q_qd=scratch)
    torch.cuda.synchronize()
    cost = self_d.view(B, H).sum(-1) + c_cost.view(B, -1).sum(-1)
    assert np.array_equal(flags.cpu().numpy(), sc["sparse_index"]), "colliding sphere pair must be identical"
    grads = (((g1 + g2[0] + gp.view(n, D)), ref_gq), ((g2[1] + gv.view(n, D)), ref_gqd), ((g2[2] + ga.view(n, D)), ref_gqdd))
    err = {"tau": _max_err(tau.cpu().numpy(), tau_ref),
           "cost_rel": float(np.abs(cost.cpu().numpy().astype(np.float64) - ref_cost).max() / np.abs(ref_cost).max()),
           "self_cost": _max_err(self_d.view(B, H).cpu().numpy(), sc["distance"].reshape(B, H)),
           "cspace_cost": _max_err(c_cost.cpu().numpy(), cs["cost"].reshape(B, H, D)),
           "grad_q": _max_err(grads[0][0].cpu().numpy(), grads[0][1]), "grad_qd": _max_err(grads[1][0].cpu().numpy(), grads[1][1]),
           "grad_qdd": _max_err(grads[2][0].cpu().numpy(), grads[2][1])}
    print(f"\n[c4 parity] B={B} H={H} rnea={'scratch' if scratch else 'staged'} max errors (fraction of the largest entry): "
          + ", ".join(f"{k} {v:.2e}" for k, v in err.items()))
    np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, rtol=1e-5, atol=1e-5 * np.abs(tau_ref).max(), err_msg=str(err))
    np.testing.assert_allclose(cost.cpu().numpy(), ref_cost, rtol=1e-5, atol=1e-6 * np.abs(ref_cost).max(), err_msg=str(err))
    for got, ref in grads:
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=5e-4, atol=5e-4 * np.abs(ref).max(), err_msg=str(err))


@pytest.mark.parametrize("robot,n", [("franka", 1000), ("unitree_g1", 300)])
def test_rnea_scratch_launches_equal_the_staged_ones(robot, n, oracle, device):
    """``curobo_hip_launch_rnea_forward_scratch`` / ``_backward_scratch`` (inputs transposed into a caller's scratch, no LDS staging):
    torques and gradients equal to the staged launches' (same walk; lanes instead of quads) and to the oracle; the variants of the
    scratch VJP (q / qd re-used from the forward's scratch, accumulation) bit-identical among themselves"""
    from curobo_amd.backends import dynamics as Dy

    model, kin, q, qd, qdd, rng = _setup(robot, device, n, 5)
    L, D = kin.num_links, kin.num_dof
    grav = np.array([0, 0, 0, 0, 0, 9.81], np.float32)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    args = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
            kin.joint_offset_map, t(grav), kin.link_level_offsets, kin.link_level_data)
    w = rng.normal(size=(n, D)).astype(np.float32)
    out = []
    for scratch in (None, torch.full((3 * n * D,), float("nan"), device=device)):
        tau, cache = torch.zeros(n, D, device=device), torch.zeros(n, L * 20, device=device)
        Dy.launch_rnea_forward(tau, t(q), t(qd), t(qdd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=scratch)
        g = [torch.full((n, D), 7.0, device=device) for _ in range(3)]
        Dy.launch_rnea_backward(*g, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=scratch)
        torch.cuda.synchronize()
        out.append((tau, cache, g))
        if scratch is not None:  # q and qd of the forward launch are still in the scratch: only grad_tau is transposed again
            g2 = [torch.full((n, D), 7.0, device=device) for _ in range(3)]
            Dy.launch_rnea_forward(tau, t(q), t(qd), t(qdd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=scratch)
            Dy.launch_rnea_backward(*g2, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=scratch,
                                    scratch_holds_q_qd=True)
            torch.cuda.synchronize()
            for a, b in zip(g, g2):
                assert torch.equal(a, b)
            # accumulate: added to what the buffers hold
            g3 = [torch.full((n, D), 0.5, device=device) for _ in range(3)]
            Dy.launch_rnea_backward(*g3, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=scratch,
                                    scratch_holds_q_qd=True, accumulate=True)
            torch.cuda.synchronize()
            for a, b in zip(g, g3):
                torch.testing.assert_close(b, a + 0.5, rtol=1e-6, atol=1e-6 * float(a.abs().max()))
            with pytest.raises(ValueError, match="accumulate needs"):
                Dy.launch_rnea_backward(*g3, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, accumulate=True)
    (tau0, cache0, g0), (tau1, cache1, g1) = out
    # (the staged launches walk on quads, the scratch launches on lanes by default: same arithmetic, another summation order)
    torch.testing.assert_close(tau0, tau1, rtol=1e-5, atol=1e-5 * float(tau0.abs().max()))
    for a, b in zip(g0, g1):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()))
    tau_ref, _ = oracle.rnea_forward(q, qd, qdd, model.as_dict(), gravity=grav)
    np.testing.assert_allclose(tau1.cpu().numpy(), tau_ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(tau_ref).max()))
    with pytest.raises(ValueError, match="scratch must hold"):
        Dy.launch_rnea_forward(tau, t(q), t(qd), t(qdd), *args, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=torch.zeros(8, device=device))
