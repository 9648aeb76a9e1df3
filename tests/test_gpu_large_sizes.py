"""Maximum sizes: batches whose flat element index passes 2^31 (a [points, spheres, 4] float buffer of 8.7 GB) through FK,
the FK VJP, self collision, swept scene collision and the per-trajectory cost sum; whole trajectories sampled at the start, the
end, around the 2^31-element boundary and at random are compared with the oracle.  (288 GB of HBM make such batches ordinary:
the reference's planners size their batch by problems x seeds x horizon, never by an index width.)"""
import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def test_batches_past_two_to_the_31_elements(oracle, device):
    from curobo_amd.backends import collision as Cn
    from curobo_amd.backends import geometry as G
    from curobo_amd.backends import kinematics as K
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays

    free, _ = torch.cuda.mem_get_info(device)
    if free < 64 << 30:
        pytest.skip("needs 64 GB of free device memory")
    model = load_model("franka")
    kp = KinematicsParams.from_model(model, device)
    d, S, L, T = model.num_dof, model.num_spheres, model.num_links, kp.num_pose_links
    P = model.collision_pairs.shape[0]
    b, h = 400_001, 21
    n = b * h
    assert n * S * 4 > 2**31
    gen = torch.Generator(device=device).manual_seed(5)
    lo, hi = (torch.as_tensor(np.asarray(v, np.float32), device=device) for v in model.joint_limits_position)
    lo, hi = lo.float() * 0.6, hi.float() * 0.6
    q0 = lo + (hi - lo) * torch.rand(b, 1, d, device=device, generator=gen)
    q1 = lo + (hi - lo) * torch.rand(b, 1, d, device=device, generator=gen)
    tt = torch.linspace(0, 1, h, device=device).view(1, h, 1)
    q = (q0 * (1 - tt) + q1 * tt).reshape(n, d).contiguous()
    del q0, q1

    # ---- FK over every point
    link_pos, link_quat = torch.zeros(n, T, 3, device=device), torch.zeros(n, T, 4, device=device)
    spheres, com, cumul = torch.zeros(n, S, 4, device=device), torch.zeros(n, 4, device=device), torch.zeros(n, L, 3, 4, device=device)
    env_b = torch.zeros(b, dtype=torch.int32, device=device)
    K.launch_kinematics_forward_spheres(link_pos, link_quat, spheres, com, cumul, q, kp.fixed_transforms, kp.link_spheres, kp.link_masses_com,
                                        kp.joint_map_type, kp.joint_map, kp.link_map, kp.tool_frame_map, kp.link_sphere_idx_map,
                                        kp.joint_offset_map, env_b, kp.num_envs, n, h, d, S, 32, True, True)
    # ---- self collision over every point
    self_d = torch.full((n, 1), -1.0, device=device)
    self_g = torch.zeros(n, S, 4, device=device)
    flags = torch.zeros(n, S, dtype=torch.uint8, device=device)
    G.self_collision_distance(self_d, self_g, torch.zeros(1, device=device), flags, spheres, kp.self_collision.sphere_padding,
                              torch.tensor([2.5], device=device), kp.self_collision.collision_pairs, torch.zeros(1, device=device),
                              torch.zeros(2, dtype=torch.int16, device=device), 1, 256, n, 1, S, P, False, True)
    # ---- swept scene collision with the speed metric over every trajectory
    c, s = np.cos(0.4), np.sin(0.4)
    arrays = cuboid_scene_arrays([[
        {"dims": [2.2, 2.2, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]},
        {"dims": [0.1, 0.1, 1.5], "pose": [0.45, 0.0, 0.3, 1, 0, 0, 0]},
        {"dims": [0.3, 0.4, 0.5], "pose": [0.3, 0.5, 0.4, c, 0, 0, s]},
        {"dims": [0.5, 0.1, 0.6], "pose": [0.1, -0.5, 0.5, c, 0, s, 0]},
    ]])
    scene = SceneData.from_arrays(arrays, device)
    w, eta = 3.0, 0.03
    sc_d, sc_g = torch.full((b, h, S), 5.0, device=device), torch.full((b, h, S, 4), 5.0, device=device)
    Cn.sphere_obstacle_collision(sc_d, sc_g, spheres.view(b, h, S, 4), scene.struct, torch.tensor([w], device=device),
                                 torch.tensor([eta], device=device), None, b, h, S, False, 3, True, torch.tensor([0.05], device=device))
    # ---- per-trajectory cost sum, FK VJP of the scene gradient
    total = torch.zeros(b, device=device)
    Cn.trajectory_cost_sum(total, self_d.view(b, h), sc_d, b, h, S)
    grad_q = torch.zeros(n, d, device=device)
    zp, zq, z4 = torch.zeros(n, T, 3, device=device), torch.zeros(n, T, 4, device=device), torch.zeros(n, 4, device=device)
    K.launch_kinematics_backward(grad_q, zp, zq, sc_g.view(n, S, 4), z4, com, zp, cumul, kp.link_spheres, kp.link_masses_com, kp.link_map,
                                 kp.joint_map, kp.joint_map_type, kp.tool_frame_map, kp.link_sphere_idx_map, kp.link_chain_data,
                                 kp.link_chain_offsets, kp.joint_links_data, kp.joint_links_offsets, kp.joint_affects_endeffector,
                                 kp.joint_offset_map, torch.zeros(n, dtype=torch.int32, device=device), kp.num_envs, n, 1, d, S, True, False)
    torch.cuda.synchronize()

    # ---- the sampled trajectories against the oracle
    edge = 2**31 // (h * S * 4)  # the trajectory holding flat float index 2^31 of the sphere buffer
    rng = np.random.default_rng(3)
    rows = np.unique(np.concatenate([np.arange(3), np.arange(b - 3, b), np.arange(edge - 3, edge + 4),
                                     [2**31 // (h * L * 12), 2**31 // (h * S)], rng.integers(0, b, size=24)])).astype(np.int64)
    rows = rows[rows < b]
    ridx = torch.as_tensor(rows, device=device)
    k = rows.size
    qs = q.view(b, h, d)[ridx].cpu().numpy().reshape(k * h, d)
    ref = oracle.kinematics_forward(qs, model.as_dict(), compute_com=True, horizon=h)
    np.testing.assert_allclose(spheres.view(b, h, S, 4)[ridx].cpu().numpy().reshape(k * h, S, 4), ref["robot_spheres"].reshape(k * h, S, 4), atol=1e-5, rtol=0)
    np.testing.assert_allclose(link_pos.view(b, h, T, 3)[ridx].cpu().numpy().reshape(k * h, T, 3), ref["link_pos"].reshape(k * h, T, 3), atol=1e-5, rtol=0)
    np.testing.assert_allclose(cumul.view(b, h, L, 3, 4)[ridx].cpu().numpy().reshape(k * h, L, 3, 4), ref["cumul_mat"].reshape(k * h, L, 3, 4), atol=1e-5, rtol=0)
    np.testing.assert_allclose(com.view(b, h, 4)[ridx].cpu().numpy().reshape(k * h, 4), ref["com"].reshape(k * h, 4), atol=1e-5, rtol=0)
    qa, qb = link_quat.view(b, h, T, 4)[ridx].cpu().numpy().reshape(k * h, T, 4), ref["link_quat"].reshape(k * h, T, 4)
    assert (np.abs(np.abs((qa * qb).sum(-1)) - 1.0) < 1e-5).all()
    # self collision on the product's own spheres (bit-identical input)
    sph_s = spheres.view(b, h, S, 4)[ridx].cpu().numpy()
    rs = oracle.self_collision(sph_s.reshape(k * h, S, 4), model.sphere_padding, model.collision_pairs, 2.5)
    np.testing.assert_allclose(self_d.view(b, h)[ridx].cpu().numpy().reshape(-1), rs["distance"], atol=2e-5, rtol=1e-5)
    assert np.array_equal(flags.view(b, h, S)[ridx].cpu().numpy().reshape(k * h, S), rs["sparse_index"])
    np.testing.assert_allclose(self_g.view(b, h, S, 4)[ridx].cpu().numpy().reshape(k * h, S, 4), rs["gradient"], atol=1e-4, rtol=1e-4)
    # scene collision
    rc = oracle.scene_collision(sph_s, arrays, w, eta, sweep=True, enable_speed_metric=True, speed_dt=0.05)
    dd, gg = sc_d[ridx].cpu().numpy(), sc_g[ridx].cpu().numpy()
    stepn = np.linalg.norm(np.diff(sph_s[..., :3], axis=1), axis=-1)
    ok = np.ones(dd.shape, bool)
    ok[:, 1:] &= stepn >= 1e-5
    ok[:, :-1] &= stepn >= 1e-5
    scl = 20.0 * w
    e = np.abs(dd - rc["distance"])[ok]
    assert int((e > 3e-5 * scl + 2e-4 * np.abs(rc["distance"])[ok]).sum()) <= 2, float(e.max())
    eg = np.abs(gg - rc["gradient"]).max(-1)[ok]
    assert int((eg > 3e-4 * scl + 2e-3 * np.abs(rc["gradient"]).max(-1)[ok]).sum()) <= 2 + int(2e-3 * (rc["distance"] > 0).sum()), float(eg.max())
    # cost sum of the product's own costs (fp32 sum of h * (S + 1) terms)
    want = self_d.view(b, h)[ridx].double().sum(1) + sc_d[ridx].double().sum((1, 2))
    np.testing.assert_allclose(total[ridx].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-5)
    # FK VJP of the product's own scene gradient on the product's own transforms
    rb = oracle.kinematics_backward(model.as_dict(), cumul.view(b, h, L, 3, 4)[ridx].cpu().numpy().reshape(k * h, L, 3, 4), gg.reshape(k * h, S, 4),
                                    np.zeros((k * h, T, 3), np.float32), np.zeros((k * h, T, 4), np.float32), np.zeros((k * h, 4), np.float32),
                                    com.view(b, h, 4)[ridx].cpu().numpy().reshape(k * h, 4))
    np.testing.assert_allclose(grad_q.view(b, h, d)[ridx].cpu().numpy().reshape(k * h, d), rb, atol=2e-5 * max(float(np.abs(rb).max()), 1.0), rtol=2e-4)
    # every trajectory was written: no cell of the outputs holds its fill value, and the whole-batch sums are finite
    assert bool((sc_d != 5.0).all()) and bool((self_d != -1.0).all())
    assert bool(torch.isfinite(total).all()) and bool(torch.isfinite(grad_q).all())
    del spheres, cumul, sc_g, self_g, sc_d, grad_q
    torch.cuda.empty_cache()


def _rows(b, per_row, rng):
    """rows of a [b, per_row]-element buffer to sample: both ends, the rows around flat element 2^31, some at random"""
    edge = 2**31 // per_row
    rows = np.unique(np.concatenate([np.arange(3), np.arange(b - 3, b), np.arange(edge - 3, edge + 4), rng.integers(0, b, size=40)]))
    return rows[(rows >= 0) & (rows < b)].astype(np.int64)


def test_lbfgs_step_with_a_history_past_two_to_the_31_elements(oracle, device):
    from curobo_amd.backends import optimization as Op

    free, _ = torch.cuda.mem_get_info(device)
    if free < 64 << 30:
        pytest.skip("needs 64 GB of free device memory")
    b, v, m = 5_200_001, 84, 5
    assert m * b * v > 2**31
    gen = torch.Generator(device=device).manual_seed(11)
    mk = lambda *s: torch.randn(*s, device=device, generator=gen)  # noqa: E731
    # a curvature history (y = D s with a positive diagonal D, rho = 1 / y.s): the two-loop recursion is then well conditioned
    dv = dict(step=torch.zeros(b, v, device=device), s=mk(m, b, v), x0=mk(b, v), g0=mk(b, v))
    dv["y"] = dv["s"] * (0.5 + torch.rand(m, b, v, device=device, generator=gen))
    dv["rho"] = 1.0 / (dv["y"] * dv["s"]).sum(-1)
    dv["q"] = dv["x0"] + 0.1 * mk(b, v)
    dv["g"] = dv["g0"] + (dv["q"] - dv["x0"]) * (0.5 + torch.rand(b, v, device=device, generator=gen))
    rng = np.random.default_rng(2)
    # the rows whose slot-(m-1) history entries straddle flat element 2^31, besides the ends and a random set
    rows = np.unique(np.concatenate([_rows(b, v, rng), np.arange(3) + (2**31 // v - (m - 1) * b)]))
    rows = rows[(rows >= 0) & (rows < b)]
    ridx = torch.as_tensor(rows, device=device)
    st = {k: (t[:, ridx] if t.shape[0] == m else t[ridx]).cpu().numpy().copy() for k, t in dv.items()}
    for _ in range(2):
        oracle.lbfgs_step(st["step"], st["rho"], st["y"], st["s"], st["q"], st["g"], st["x0"], st["g0"], 0.01, True)
        Op.launch_lbfgs_step(dv["step"], dv["rho"], dv["y"], dv["s"], dv["q"], dv["g"], dv["x0"], dv["g0"], 0.01, b, m, v, True, True)
        torch.cuda.synchronize()
        for k in ("step", "rho", "y", "s", "x0", "g0"):
            got = (dv[k][:, ridx] if dv[k].shape[0] == m else dv[k][ridx]).cpu().numpy()
            np.testing.assert_allclose(got, st[k], atol=3e-5 * max(1.0, float(np.abs(st[k]).max())), rtol=3e-4, err_msg=k)
        dv["q"].add_(dv["step"], alpha=0.01)
        st["q"] = dv["q"][ridx].cpu().numpy().copy()
        dv["g"].add_((0.01 * dv["step"]) * (0.5 + torch.rand(b, v, device=device, generator=gen)))
        st["g"] = dv["g"][ridx].cpu().numpy().copy()
    assert bool(torch.isfinite(dv["step"]).all())
    del dv
    torch.cuda.empty_cache()


def test_bspline_trajectories_past_two_to_the_31_elements(oracle, device):
    from curobo_amd.backends import trajectory as Tr

    free, _ = torch.cuda.mem_get_info(device)
    if free < 96 << 30:
        pytest.skip("needs 96 GB of free device memory")
    b, nk, dof, degree, interp = 9_400_001, 12, 7, 3, 2
    ph = (nk + degree + 1) * interp + 1
    assert b * ph * dof > 2**31
    keys = ("position", "velocity", "acceleration", "jerk")
    gen = torch.Generator(device=device).manual_seed(13)
    u = torch.randn(b, nk, dof, device=device, generator=gen)
    rng = np.random.default_rng(4)
    ns, ng = 3, 2
    mk = lambda n: {k: rng.normal(size=(n, dof)).astype(np.float32) * 0.3 for k in keys}  # noqa: E731
    start, goal = mk(ns), mk(ng)
    sidx = torch.randint(0, ns, (b,), device=device, generator=gen, dtype=torch.int32)
    gidx = torch.randint(0, ng, (b,), device=device, generator=gen, dtype=torch.int32)
    dt, imp = np.asarray([0.05, 0.11], np.float32), np.asarray([1, 0], np.uint8)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    outs = [torch.full((b, ph, dof), 7.0, device=device) for _ in range(4)]
    out_dt = torch.zeros(b, device=device)
    Tr.launch_bspline_interpolation_forward_kernel(*outs, out_dt, u, *[t(start[k]) for k in keys], *[t(goal[k]) for k in keys], sidx, gidx,
                                                   t(dt), t(imp), b, ph, dof, nk, degree)
    torch.cuda.synchronize()
    rows = _rows(b, ph * dof, rng)
    ridx = torch.as_tensor(rows, device=device)
    ref = oracle.bspline_forward(u[ridx].cpu().numpy(), start, goal, sidx[ridx].cpu().numpy(), gidx[ridx].cpu().numpy(), dt, imp, ph, degree)
    for o_, k in zip(outs, keys):
        np.testing.assert_allclose(o_[ridx].cpu().numpy(), ref[k], atol=2e-5 * max(1.0, float(np.abs(ref[k]).max())), rtol=2e-5, err_msg=k)
        assert bool((o_[-1] != 7.0).any()) and bool(torch.isfinite(o_[b // 2:]).all())
    # VJP: the four trajectories themselves as the incoming gradients
    og = torch.full((b, nk, dof), 7.0, device=device)
    Tr.launch_bspline_interpolation_backward_kernel(og, *outs, t(dt), gidx, t(imp), b, ph, dof, nk, degree, False)
    torch.cuda.synchronize()
    refb = oracle.bspline_backward(*[o_[ridx].cpu().numpy() for o_ in outs], dt, gidx[ridx].cpu().numpy(), imp, nk, degree)
    np.testing.assert_allclose(og[ridx].cpu().numpy(), refb, atol=2e-5 * max(1.0, float(np.abs(refb).max())), rtol=2e-4)
    assert bool(torch.isfinite(og).all()) and bool((og.view(b, -1) != 7.0).any(1).all())
    del outs, og, u
    torch.cuda.empty_cache()
