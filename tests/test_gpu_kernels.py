"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances: FK poses / spheres / costs 1e-5 abs (north_star), gradients
1e-4 rel-to-scale (different fp32 summation order), integer outputs exact."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q
from sweep_allowance import assert_scene_kernel_parity

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def _kp(model, device):
    from curobo_amd.robot.kinematics_params import KinematicsParams

    return KinematicsParams.from_model(model, device)


def _fk_gpu(kp, q, horizon=1, jac=False, com=False, env_query_idx=None):
    from curobo_amd.backends import kinematics as K

    dev = kp.device
    n, d = q.shape
    T, S, L = kp.num_pose_links, kp.num_spheres, kp.num_links
    qd = torch.as_tensor(q, device=dev)
    o = dict(
        link_pos=torch.zeros(n, T, 3, device=dev), link_quat=torch.zeros(n, T, 4, device=dev),
        spheres=torch.zeros(n, S, 4, device=dev), com=torch.zeros(n, 4, device=dev),
        jac=torch.zeros(n, T, 6, d, device=dev), cumul=torch.zeros(n, L, 3, 4, device=dev),
    )
    env = torch.zeros(max(n // horizon, 1), dtype=torch.int32, device=dev) if env_query_idx is None \
        else torch.as_tensor(env_query_idx, device=dev)
    if jac:
        K.launch_kinematics_forward_spheres_jacobian(
            o["link_pos"], o["link_quat"], o["spheres"], o["com"], o["jac"], o["cumul"], qd,
            kp.fixed_transforms, kp.link_spheres, kp.link_masses_com, kp.joint_map_type, kp.joint_map,
            kp.link_map, kp.tool_frame_map, kp.link_sphere_idx_map, kp.link_chain_data,
            kp.link_chain_offsets, kp.joint_links_data, kp.joint_links_offsets,
            kp.joint_affects_endeffector, kp.joint_offset_map, env, kp.num_envs, n, horizon, d, S, 32,
            True, com)
    else:
        K.launch_kinematics_forward_spheres(
            o["link_pos"], o["link_quat"], o["spheres"], o["com"], o["cumul"], qd, kp.fixed_transforms,
            kp.link_spheres, kp.link_masses_com, kp.joint_map_type, kp.joint_map, kp.link_map,
            kp.tool_frame_map, kp.link_sphere_idx_map, kp.joint_offset_map, env, kp.num_envs, n,
            horizon, d, S, 32, True, com)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}


@pytest.mark.parametrize("robot", ["franka", "ur10e", "unitree_g1"])
@pytest.mark.parametrize("n", [1, 17, 1000])
def test_fk_forward(robot, n, oracle, device):
    model = load_model(robot)
    kp = _kp(model, device)
    q = sample_q(model, n, seed=n)
    ref = oracle.kinematics_forward(q, model.as_dict(), compute_jacobian=True, compute_com=True)
    got = _fk_gpu(kp, q, jac=True, com=True)
    np.testing.assert_allclose(got["link_pos"], ref["link_pos"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got["link_quat"], ref["link_quat"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got["spheres"], ref["robot_spheres"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got["cumul"], ref["cumul_mat"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got["com"], ref["com"], atol=2e-5, rtol=1e-6)
    np.testing.assert_allclose(got["jac"], ref["jacobian"], atol=ATOL, rtol=0)
    # the spheres-only entry point runs the one-lane-per-point kernel (a different order of the same products):
    # held to the oracle like the Jacobian variant, and to the Jacobian variant within a few ulp
    got2 = _fk_gpu(kp, q, jac=False, com=False)
    for k, rk in (("link_pos", "link_pos"), ("link_quat", "link_quat"), ("spheres", "robot_spheres"), ("cumul", "cumul_mat")):
        np.testing.assert_allclose(got2[k], ref[rk], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got2["spheres"], got["spheres"], atol=2e-6, rtol=0)


def test_fk_known_answer(device):
    """reference curobo/tests/_src/robot/kinematics/test_kinematics.py:57-82"""
    model = load_model("franka")
    kp = _kp(model, device)
    got = _fk_gpu(kp, np.array([[0, -1.2, 0, -2, 0, 1, 0]], np.float32))
    np.testing.assert_allclose(got["link_pos"][0, 0], [6.0860e-02, -4.7547e-12, 7.6373e-01], atol=1e-5)
    np.testing.assert_allclose(got["link_quat"][0, 0], [0.0382, 0.9193, 0.3808, 0.0922], atol=1e-4)


@pytest.mark.parametrize("robot", ["franka", "ur10e", "unitree_g1"])
def test_fk_backward(robot, oracle, device):
    from curobo_amd.backends import kinematics as K

    model = load_model(robot)
    kp = _kp(model, device)
    n = 257
    q = sample_q(model, n, seed=3)
    fwd = oracle.kinematics_forward(q, model.as_dict(), compute_com=True)
    rng = np.random.default_rng(1)
    S, T, L, d = model.num_spheres, len(model.tool_frames), model.num_links, model.num_dof
    g_s = rng.normal(size=(n, S, 4)).astype(np.float32)
    g_s[rng.uniform(size=(n, S)) < 0.6] = 0.0  # sparse, like collision gradients
    g_b = rng.normal(size=(n, S, 4)).astype(np.float32)
    g_b[rng.uniform(size=(n, S)) < 0.8] = 0.0
    g_p = rng.normal(size=(n, T, 3)).astype(np.float32)
    g_q = rng.normal(size=(n, T, 4)).astype(np.float32)
    g_c = rng.normal(size=(n, 4)).astype(np.float32)
    ref = oracle.kinematics_backward(model.as_dict(), fwd["cumul_mat"], g_s + g_b * np.array([1, 1, 1, 0], np.float32),
                                     g_p, g_q, g_c, fwd["com"])
    dev = device
    out = torch.zeros(n, d, device=dev)
    env = torch.zeros(n, dtype=torch.int32, device=dev)
    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    K.launch_kinematics_backward(
        out, t(g_p), t(g_q), t(g_s), t(g_c), t(fwd["com"]), t(g_p), t(fwd["cumul_mat"]), kp.link_spheres,
        kp.link_masses_com, kp.link_map, kp.joint_map, kp.joint_map_type, kp.tool_frame_map,
        kp.link_sphere_idx_map, kp.link_chain_data, kp.link_chain_offsets, kp.joint_links_data,
        kp.joint_links_offsets, kp.joint_affects_endeffector, kp.joint_offset_map, env, kp.num_envs, n, 1,
        d, S, True, False, grad_spheres_b=t(g_b))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got, ref, atol=1e-5 * max(scale, 1.0), rtol=1e-4)


@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_fk_sphere_sets_per_environment(robot, oracle, device):
    """two sets of robot spheres selected per batch row through env_query_idx (reference
    kinematics_forward_helper.cuh:218-254 / kinematics_backward_kernel.cuh:48-52: attached objects differ per
    environment): forward spheres (both entry points) and the sphere VJP against the oracle run once per set"""
    import dataclasses

    from curobo_amd.backends import kinematics as K

    model = load_model(robot)
    kp = _kp(model, device)
    rng = np.random.default_rng(11)
    S, d = model.num_spheres, model.num_dof
    base = np.asarray(model.as_dict()["link_spheres"], np.float32).reshape(-1, S, 4)[0]
    sph2 = np.array(base, copy=True)
    sph2[:, :3] += rng.normal(scale=0.02, size=(S, 3)).astype(np.float32)
    sph2[:, 3] = np.abs(sph2[:, 3]) * 1.5
    both = np.stack([base, sph2])
    kp2 = dataclasses.replace(kp, link_spheres=torch.as_tensor(both, device=device))
    batch, horizon = 70, 3  # 210 points: not a multiple of the 64-point workgroup, rows of both sets in one workgroup
    n = batch * horizon
    q = sample_q(model, n, seed=5)
    env = (np.arange(batch) % 2).astype(np.int32)
    refs = []
    for e in range(2):
        md = dict(model.as_dict())
        md["link_spheres"] = both[e:e + 1]
        refs.append(oracle.kinematics_forward(q, md))
    row_env = np.repeat(env, horizon)
    want = np.where(row_env[:, None, None] == 0, refs[0]["robot_spheres"], refs[1]["robot_spheres"])
    for jac in (False, True):
        got = _fk_gpu(kp2, q, horizon=horizon, jac=jac, com=False, env_query_idx=env)
        np.testing.assert_allclose(got["spheres"], want, atol=ATOL, rtol=0)
    # VJP: the sphere centres in world frame depend on the set
    g_s = rng.normal(size=(n, S, 4)).astype(np.float32)
    g_s[rng.uniform(size=(n, S)) < 0.5] = 0.0
    T = len(model.tool_frames)
    zp, zq, zc = np.zeros((n, T, 3), np.float32), np.zeros((n, T, 4), np.float32), np.zeros((n, 4), np.float32)
    ref_g = []
    for e in range(2):
        md = dict(model.as_dict())
        md["link_spheres"] = both[e:e + 1]
        ref_g.append(oracle.kinematics_backward(md, refs[e]["cumul_mat"], g_s, zp, zq))
    want_g = np.where(row_env[:, None] == 0, ref_g[0], ref_g[1])
    out = torch.zeros(n, d, device=device)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    K.launch_kinematics_backward(
        out, t(zp), t(zq), t(g_s), t(zc), t(zc), t(zp), t(refs[0]["cumul_mat"]), kp2.link_spheres,
        kp2.link_masses_com, kp2.link_map, kp2.joint_map, kp2.joint_map_type, kp2.tool_frame_map,
        kp2.link_sphere_idx_map, kp2.link_chain_data, kp2.link_chain_offsets, kp2.joint_links_data,
        kp2.joint_links_offsets, kp2.joint_affects_endeffector, kp2.joint_offset_map, t(env), 2, n, horizon,
        d, S, False, False)
    torch.cuda.synchronize()
    scale = max(np.abs(want_g).max(), 1.0)
    np.testing.assert_allclose(out.cpu().numpy(), want_g, atol=1e-5 * scale, rtol=1e-4)


@pytest.mark.parametrize("robot,n", [("franka", 515), ("ur10e", 64), ("unitree_g1", 33)])
def test_self_collision(robot, n, oracle, device):
    from curobo_amd.backends import geometry as G

    model = load_model(robot)
    kp = _kp(model, device)
    q = sample_q(model, n, seed=5)
    sph = oracle.kinematics_forward(q, model.as_dict())["robot_spheres"]
    S, P = model.num_spheres, model.collision_pairs.shape[0]
    dev = device
    # stale state from a "previous call": flagged rows must be zeroed
    stale_grad = np.zeros((n, S, 4), np.float32)
    stale_flag = np.zeros((n, S), np.uint8)
    stale_grad[:, 3] = 7.0
    stale_flag[:, 3] = 1
    ref = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 2.5,
                                out_gradient=stale_grad.copy(), sparse_index=stale_flag.copy(),
                                store_pair_distance=True)
    out_d = torch.full((n, 1), -1.0, device=dev)
    out_g = torch.as_tensor(stale_grad, device=dev)
    flags = torch.as_tensor(stale_flag, device=dev)
    pd = torch.zeros(n, P, device=dev)
    w = torch.tensor([2.5], device=dev)
    G.self_collision_distance(
        out_d, out_g, pd, flags, torch.as_tensor(sph, device=dev), kp.self_collision.sphere_padding, w,
        kp.self_collision.collision_pairs, torch.zeros(1, device=dev), torch.zeros(2, dtype=torch.int16, device=dev),
        1, 256, n, 1, S, P, True, True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_d.cpu().numpy()[:, 0], ref["distance"], atol=ATOL, rtol=1e-5)
    assert np.array_equal(flags.cpu().numpy(), ref["sparse_index"]), "collision-pair indices must be exact"
    np.testing.assert_allclose(out_g.cpu().numpy(), ref["gradient"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(pd.cpu().numpy(), ref["pair_distance"], atol=ATOL, rtol=1e-5)
    assert (ref["distance"] > 0).any(), "test inputs must contain self collisions"


def _scene_arrays():
    from curobo_amd.scene import cuboid_scene_arrays

    c, s = np.cos(0.4), np.sin(0.4)
    return cuboid_scene_arrays([[
        {"dims": [2.2, 2.2, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]},
        {"dims": [0.1, 0.1, 1.5], "pose": [0.45, 0.0, 0.3, 1, 0, 0, 0]},
        {"dims": [0.3, 0.4, 0.5], "pose": [0.3, 0.5, 0.4, c, 0, 0, s]},
        {"dims": [0.2, 0.2, 0.2], "pose": [-0.4, -0.3, 0.6, c, s, 0, 0], "enable": False},
        {"dims": [0.5, 0.1, 0.6], "pose": [0.1, -0.5, 0.5, c, 0, s, 0]},
    ]])


@pytest.mark.parametrize("sweep,speed", [(False, False), (True, False), (True, True)])
def test_scene_collision_cuboids(sweep, speed, oracle, device):
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    model = load_model("franka")
    b, h = 24, 9
    rng = np.random.default_rng(9)
    q0 = sample_q(model, b, seed=11)[:, None, :]
    q1 = sample_q(model, b, seed=12)[:, None, :]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    q = (q0 * (1 - tt) + q1 * tt).reshape(b * h, -1) * 0.6
    sph = oracle.kinematics_forward(q, model.as_dict(), horizon=h)["robot_spheres"].reshape(b, h, -1, 4)
    arrays = _scene_arrays()
    ref = oracle.scene_collision(sph, arrays, 3.0, 0.03, sweep=sweep, enable_speed_metric=speed, speed_dt=0.05)
    scene = SceneData.from_arrays(arrays, device)
    S = sph.shape[2]
    dist = torch.full((b, h, S), 5.0, device=device)
    grad = torch.full((b, h, S, 4), 5.0, device=device)
    Cn.sphere_obstacle_collision(
        dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([3.0], device=device),
        torch.tensor([0.03], device=device), torch.zeros(b, dtype=torch.int32, device=device), b, h, S,
        False, 3 if sweep else 0, speed, torch.tensor([0.05], device=device))
    torch.cuda.synchronize()
    assert (ref["distance"] > 0).mean() > 0.02
    assert_scene_kernel_parity(oracle, dist.cpu().numpy(), grad.cpu().numpy(), sph, arrays, 3.0, 0.03, f"cuboids sweep={sweep} speed={speed}",
                               sweep=sweep, enable_speed_metric=speed, speed_dt=0.05)


def test_scene_collision_voxels(oracle, device):
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, voxel_grid_from_sdf

    def sdf(p):  # union of a box and a sphere
        qb = np.abs(p - np.array([0.4, 0.0, 0.3])) - np.array([0.15, 0.3, 0.3])
        box = np.linalg.norm(np.maximum(qb, 0), axis=-1) + np.minimum(qb.max(-1), 0)
        sp = np.linalg.norm(p - np.array([-0.3, 0.4, 0.5]), axis=-1) - 0.2
        return np.minimum(box, sp)

    model = load_model("ur10e")
    arrays = voxel_grid_from_sdf(sdf, (48, 40, 44), 0.04, pose7=(0.05, -0.02, 0.4, 1, 0, 0, 0), max_distance=100.0)
    b, h = 16, 8
    q = sample_q(model, b * h, seed=21)
    sph = oracle.kinematics_forward(q, model.as_dict(), horizon=h)["robot_spheres"].reshape(b, h, -1, 4)
    sph[0, 0, 0, :3] = [5.0, 5.0, 5.0]  # far outside the grid: boundary path
    sph[0, 0, 1, :3] = [0.05 + 0.95, -0.02, 0.4]  # straddles the grid face
    ref = oracle.scene_collision(sph, arrays, 1.0, 0.05, sweep=True)
    scene = SceneData.from_arrays(arrays, device)
    S = sph.shape[2]
    dist = torch.zeros(b, h, S, device=device)
    grad = torch.zeros(b, h, S, 4, device=device)
    Cn.sphere_obstacle_collision(
        dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
        torch.tensor([0.05], device=device), None, b, h, S, False, 3, False, None)
    torch.cuda.synchronize()
    assert (ref["distance"] > 0).mean() > 0.02
    assert_scene_kernel_parity(oracle, dist.cpu().numpy(), grad.cpu().numpy(), sph, arrays, 1.0, 0.05, "voxels swept", voxel=True, sweep=True)


@pytest.mark.parametrize("degree", [3, 4, 5])
@pytest.mark.parametrize("implicit", [False, True])
def test_bspline(degree, implicit, oracle, device):
    from curobo_amd.backends import trajectory as Tr

    rng = np.random.default_rng(degree)
    b, nk, dof, interp = 19, 12, 7, 2
    ph = (nk + degree + 1) * interp + 1
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda n: {k: rng.normal(size=(n, dof)).astype(np.float32) * 0.3  # noqa: E731
                    for k in ("position", "velocity", "acceleration", "jerk")}
    start, goal = mk(3), mk(2)
    sidx = rng.integers(0, 3, size=b).astype(np.int32)
    gidx = rng.integers(0, 2, size=b).astype(np.int32)
    dt = np.array([0.05, 0.08], np.float32)
    imp = np.array([implicit, implicit], np.uint8)
    ref = oracle.bspline_forward(u, start, goal, sidx, gidx, dt, imp, ph, degree)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    outs = [torch.zeros(b, ph, dof, device=device) for _ in range(4)]
    out_dt = torch.zeros(b, device=device)
    keys = ("position", "velocity", "acceleration", "jerk")
    Tr.launch_bspline_interpolation_forward_kernel(
        *outs, out_dt, t(u), *[t(start[k]) for k in keys], *[t(goal[k]) for k in keys], t(sidx), t(gidx),
        t(dt), t(imp), b, ph, dof, nk, degree)
    torch.cuda.synchronize()
    for o, k in zip(outs, keys):
        scale = max(1.0, np.abs(ref[k]).max())
        np.testing.assert_allclose(o.cpu().numpy(), ref[k], atol=1e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(out_dt.cpu().numpy(), ref["dt"])
    g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
    refb = oracle.bspline_backward(*g, dt, gidx, imp, nk, degree)
    og = torch.zeros(b, nk, dof, device=device)
    Tr.launch_bspline_interpolation_backward_kernel(og, *[t(x) for x in g], t(dt), t(gidx), t(imp), b, ph,
                                                    dof, nk, degree, False)
    torch.cuda.synchronize()
    scale = max(1.0, np.abs(refb).max())
    np.testing.assert_allclose(og.cpu().numpy(), refb, atol=1e-5 * scale, rtol=1e-4)


@pytest.mark.parametrize("v_dim,m", [(7, 7), (84, 27), (84, 0), (300, 15), (1000, 5)])
def test_lbfgs_step(v_dim, m, oracle, device):
    from curobo_amd.backends import optimization as Op

    rng = np.random.default_rng(v_dim + m)
    b = 37
    mk = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
    st = dict(step=np.zeros((b, v_dim), np.float32), rho=mk(m, b) * 0.1, y=mk(m, b, v_dim), s=mk(m, b, v_dim),
              q=mk(b, v_dim), g=mk(b, v_dim), x0=mk(b, v_dim), g0=mk(b, v_dim))
    st["rho"][:, 0] = 0.0
    dv = {k: torch.as_tensor(v.copy(), device=device) for k, v in st.items()}
    for it in range(3):  # several consecutive steps exercise the history roll
        oracle.lbfgs_step(st["step"], st["rho"], st["y"], st["s"], st["q"], st["g"], st["x0"], st["g0"], 0.01, True)
        Op.launch_lbfgs_step(dv["step"], dv["rho"], dv["y"], dv["s"], dv["q"], dv["g"], dv["x0"], dv["g0"],
                             0.01, b, m, v_dim, True, True)
        torch.cuda.synchronize()
        for k in ("step", "rho", "y", "s", "x0", "g0"):
            scale = max(1.0, float(np.abs(st[k]).max())) if st[k].size else 1.0
            np.testing.assert_allclose(dv[k].cpu().numpy(), st[k], atol=2e-5 * scale, rtol=2e-4, err_msg=f"{k} it{it}")
        st["q"] = st["q"] + 0.1 * st["step"] / max(1.0, float(np.abs(st["step"]).max()))
        st["g"] = mk(b, v_dim)
        dv["q"] = torch.as_tensor(st["q"], device=device)
        dv["g"] = torch.as_tensor(st["g"], device=device)


@pytest.mark.parametrize("strong,approx", [(False, True), (True, False), (False, False)])
def test_line_search(strong, approx, oracle, device):
    from curobo_amd.backends import optimization as Op

    rng = np.random.default_rng(4)
    b, nls, v = 301, 4, 84
    x = rng.normal(size=(b, 1, v)).astype(np.float32)
    d = rng.normal(size=(b, 1, v)).astype(np.float32)
    alphas = np.array([0.0, 0.1, 0.5, 1.0], np.float32)
    sa = (x + alphas[None, :, None] * d).astype(np.float32)
    # quadratic bowl with random curvature + noise so that all branches (none / armijo-only / both) occur
    curv = rng.uniform(0.1, 3.0, size=(b, 1, v)).astype(np.float32)
    sg = (curv * sa + 0.3 * rng.normal(size=sa.shape)).astype(np.float32)
    scost = (0.5 * (curv * sa * sa).sum(-1, keepdims=True) + rng.normal(size=(b, nls, 1))).astype(np.float32)

    def fresh():
        return dict(
            best_cost=np.full((b,), 1e3, np.float32) * rng.uniform(0, 1, size=b).astype(np.float32),
            best_action=np.zeros((b, v), np.float32), best_iteration=np.zeros((b,), np.int16),
            current_iteration=rng.integers(0, 30, size=b).astype(np.int16), converged=np.zeros((b,), np.uint8),
            exploration_cost=np.zeros((b,), np.float32), exploration_action=np.zeros((b, v), np.float32),
            exploration_gradient=np.zeros((b, v), np.float32), cost=np.zeros((b,), np.float32),
            action=np.zeros((b, v), np.float32), gradient=np.zeros((b, v), np.float32),
            exploration_idx=np.zeros((b, nls), np.int32), selected_idx=np.zeros((b, nls), np.int32))

    st = fresh()
    dv = {k: torch.as_tensor(a.copy(), device=device) for k, a in st.items()}
    oracle.line_search(st, scost, sa, sg, d, alphas, 1e-5, 0.9, strong, approx, 5, 1e-4, 1e-3)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    Op.launch_line_search(
        dv["best_cost"], dv["best_action"], dv["best_iteration"], dv["current_iteration"], dv["converged"], 5,
        1e-4, 1e-3, dv["exploration_cost"], dv["exploration_action"], dv["exploration_gradient"],
        dv["exploration_idx"], dv["cost"], dv["action"], dv["gradient"], dv["selected_idx"], t(scost), t(sa), t(sg),
        t(d), t(alphas), 1e-5, 0.9, strong, approx, nls, v, b)
    torch.cuda.synchronize()
    sel = st["selected_idx"][:, 0]
    assert len(np.unique(sel)) >= 3, "inputs should exercise several line-search outcomes"
    for k in ("selected_idx", "exploration_idx", "best_iteration", "current_iteration", "converged"):
        assert np.array_equal(dv[k].cpu().numpy(), st[k]), f"{k}: integer outputs must be exact"
    for k in ("best_cost", "best_action", "exploration_cost", "exploration_action", "exploration_gradient",
              "cost", "action", "gradient"):
        np.testing.assert_array_equal(dv[k].cpu().numpy(), st[k], err_msg=k)


def test_trajectory_cost_sum(oracle, device):
    from curobo_amd.backends import collision as Cn

    rng = np.random.default_rng(0)
    b, h, S = 70, 33, 65
    sc = rng.uniform(size=(b, h, S)).astype(np.float32)
    se = rng.uniform(size=(b, h)).astype(np.float32)
    out = torch.zeros(b, device=device)
    Cn.trajectory_cost_sum(out, torch.as_tensor(se, device=device), torch.as_tensor(sc, device=device), b, h, S)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), oracle.trajectory_cost_sum(se, sc), rtol=2e-6)


@pytest.mark.parametrize("degree", [3, 4, 5])
def test_bspline_single_dt(degree, oracle, device):
    from curobo_amd.backends import trajectory as Tr

    rng = np.random.default_rng(40 + degree)
    b, nk, dof, max_out = 9, 12, 7, 150
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda n: {k: rng.normal(size=(n, dof)).astype(np.float32) * 0.3  # noqa: E731
                    for k in ("position", "velocity", "acceleration", "jerk")}
    start, goal = mk(3), mk(2)
    sidx = rng.integers(0, 3, size=b).astype(np.int32)
    gidx = rng.integers(0, 2, size=b).astype(np.int32)
    horizons = rng.integers(20, 200, size=b).astype(np.int32)  # some beyond max_out: clamped
    dt = np.array([0.02], np.float32)
    imp = np.array([0, 1], np.uint8)
    ref = oracle.bspline_single_dt(u, start, goal, sidx, gidx, dt, imp, horizons, max_out, degree)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    outs = [torch.zeros(b, max_out, dof, device=device) for _ in range(4)]
    out_dt = torch.zeros(b, device=device)
    keys = ("position", "velocity", "acceleration", "jerk")
    Tr.launch_bspline_interpolation_single_dt_kernel(
        *outs, out_dt, t(u), torch.zeros(b, nk - 1, device=device), *[t(start[k]) for k in keys],
        *[t(goal[k]) for k in keys], t(sidx), t(gidx), t(dt), t(imp), t(horizons), b, max_out, dof, nk, degree)
    torch.cuda.synchronize()
    for o, k in zip(outs, keys):
        scale = max(1.0, np.abs(ref[k]).max())
        np.testing.assert_allclose(o.cpu().numpy(), ref[k], atol=2e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(out_dt.cpu().numpy(), ref["dt"])


@pytest.mark.parametrize("use_goal", [0, 1])
def test_legacy_position_transition(use_goal, oracle, device):
    from curobo_amd.backends import trajectory as Tr

    rng = np.random.default_rng(50 + use_goal)
    b, horizon, dof = 11, 30, 7
    u = rng.normal(size=(b, horizon - 4, dof)).astype(np.float32)
    start = {k: rng.normal(size=(3, dof)).astype(np.float32) * 0.3 for k in ("position", "velocity", "acceleration")}
    goal = rng.normal(size=(2, dof)).astype(np.float32)
    sidx = rng.integers(0, 3, size=b).astype(np.int32)
    gidx = rng.integers(0, 2, size=b).astype(np.int32)
    dt = np.array([0.05, 0.08], np.float32)
    imp = np.array([use_goal, use_goal], np.uint8)
    ref = oracle.differentiation_position_forward(u, start, goal, sidx, gidx, dt, imp)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    outs = [torch.zeros(b, horizon, dof, device=device) for _ in range(4)]
    out_dt = torch.zeros(b, device=device)
    Tr.launch_differentiation_position_forward_kernel(
        *outs, out_dt, t(u), t(start["position"]), t(start["velocity"]), t(start["acceleration"]), t(goal),
        torch.zeros(2, dof, device=device), torch.zeros(2, dof, device=device), t(sidx), t(gidx), t(dt), t(imp),
        b, horizon, dof)
    torch.cuda.synchronize()
    for o, k in zip(outs, ("position", "velocity", "acceleration", "jerk")):
        scale = max(1.0, np.abs(ref[k]).max())
        np.testing.assert_allclose(o.cpu().numpy(), ref[k], atol=2e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(out_dt.cpu().numpy(), ref["dt"])
    g = [rng.normal(size=(b, horizon, dof)).astype(np.float32) for _ in range(4)]
    refb = oracle.differentiation_position_backward(*g, dt, gidx, imp)
    og = torch.zeros(b, horizon - 4, dof, device=device)
    Tr.launch_differentiation_position_backward_kernel(og, *[t(x) for x in g], t(dt), t(gidx), t(imp), b, horizon, dof)
    torch.cuda.synchronize()
    np.testing.assert_allclose(og.cpu().numpy(), refb, rtol=2e-5, atol=2e-5 * np.abs(refb).max())
    with pytest.raises(ValueError, match="horizon"):
        Tr.launch_differentiation_position_backward_kernel(og, *[t(x) for x in g], t(dt), t(gidx), t(imp), b, 8, dof)


def test_legacy_acceleration_integration(oracle, device):
    from curobo_amd.backends import trajectory as Tr

    rng = np.random.default_rng(60)
    b, horizon, dof = 13, 41, 6
    u = rng.normal(size=(b, horizon, dof)).astype(np.float32)
    start = {k: rng.normal(size=(2, dof)).astype(np.float32) for k in ("position", "velocity", "acceleration")}
    sidx = rng.integers(0, 2, size=b).astype(np.int32)
    dt = rng.uniform(0.01, 0.1, size=horizon).astype(np.float32)
    ref = oracle.integration_acceleration(u, start, sidx, dt)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    outs = [torch.zeros(b, horizon, dof, device=device) for _ in range(4)]
    Tr.launch_integration_acceleration_kernel(*outs, t(u), t(start["position"]), t(start["velocity"]),
                                              t(start["acceleration"]), t(sidx), t(dt), b, horizon, dof)
    torch.cuda.synchronize()
    for o, k in zip(outs, ("position", "velocity", "acceleration", "jerk")):
        scale = max(1.0, np.abs(ref[k]).max())
        np.testing.assert_allclose(o.cpu().numpy(), ref[k], atol=1e-5 * scale, rtol=1e-5)


@pytest.mark.parametrize("dof,n_res", [(7, 13), (6, 12), (16, 16), (17, 30), (33, 40), (49, 73), (64, 70)])
def test_levenberg_marquardt_step_mfma(dof, n_res, oracle, device):
    """J^T J on the matrix cores + in-LDS Cholesky vs the oracle (itself checked against
    numpy.linalg.solve); every tile configuration (1..4 tiles per side, ragged edges)"""
    from curobo_amd.backends import linalg as La

    rng = np.random.default_rng(dof)
    b = 37
    J = rng.normal(size=(b, n_res, dof)).astype(np.float32)
    g = np.einsum("brd,br->bd", J, rng.normal(size=(b, n_res))).astype(np.float32)
    lam = rng.uniform(1e-2, 1.0, size=b).astype(np.float32)
    q = rng.normal(size=(b, dof)).astype(np.float32)
    q_ref, pred_ref = oracle.lm_step(J, g, lam, q)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    q_out = torch.zeros(b, dof, device=device)
    pred = torch.zeros(b, device=device)
    La.levenberg_marquardt_step(q_out, pred, t(J), t(g), t(lam), t(q))
    torch.cuda.synchronize()
    d_ref = q_ref - q
    J64, g64 = J.astype(np.float64), g.astype(np.float64)
    A = np.einsum("brd,bre->bde", J64, J64) + lam[:, None, None].astype(np.float64) * np.eye(dof)
    delta = np.linalg.solve(A, -g64[..., None])[..., 0]
    # A linear solve is not a 1e-5 operation in fp32: the error of ANY fp32 solution is ~ cond(J^T J + lambda I) x 6e-8 (random
    # J with n_res ~ dof: cond up to ~1e4).  The criterion is therefore relative to the exact (fp64) solution, per problem: the
    # HIP kernel may be no further from it than 3 x the fp32 oracle is (+ 1e-5 of the step), whatever its summation order.
    scale = np.abs(delta).max(axis=1)
    e_dev = np.abs(q_out.cpu().numpy() - q - delta).max(axis=1) / scale
    e_orc = np.abs(d_ref - delta).max(axis=1) / scale
    cond = np.linalg.cond(A)
    print(f"\n[lm {dof}x{n_res}] error against the fp64 solution / step scale: HIP max {e_dev.max():.2e}, oracle max {e_orc.max():.2e}; "
          f"cond(J^T J + lambda I) up to {cond.max():.1e}; HIP vs oracle max {np.abs(q_out.cpu().numpy() - q - d_ref).max() / scale.max():.2e}")
    assert (e_dev <= 3.0 * e_orc + 1e-5).all(), (float((e_dev / (3.0 * e_orc + 1e-5)).max()), float(cond.max()))
    assert (e_dev <= 6e-8 * cond * 8 + 1e-6).all(), "within the fp32 conditioning bound of the system"
    np.testing.assert_allclose(q_out.cpu().numpy() - q, d_ref, rtol=2e-3, atol=5e-4 * np.abs(d_ref).max())
    np.testing.assert_allclose(pred.cpu().numpy(), pred_ref, rtol=2e-3, atol=1e-4 * np.abs(pred_ref).max())


def test_self_collision_dense_bitmap_kernel_c4_size(oracle, device):
    """BASELINE config 4 at full size: Unitree G1 (674 spheres, 162 111 pairs), 1024 seeds x 8 points through the
    register-tiled bitmap kernel (the drop-in entry point picks it for dense pair sets) vs the oracle and vs the
    LDS pair-list kernel: exact colliding-sphere flags, stale gradient rows cleared, disabled spheres ignored."""
    from curobo_amd._lib import check, current_stream, load, ptr
    from curobo_amd.backends import geometry as G

    model = load_model("unitree_g1")
    kp = _kp(model, device)
    B, H = 1024, 8
    n = B * H
    S, P = model.num_spheres, model.collision_pairs.shape[0]
    q = sample_q(model, n, seed=17, scale=0.6)
    sph = oracle.kinematics_forward(q, model.as_dict())["robot_spheres"].copy()
    sph[5, 100, 3] = -0.5   # a disabled sphere (negative radius): its pairs must not count
    sph[6, 0:40, 3] = -1.0
    stale_grad = np.zeros((n, S, 4), np.float32)
    stale_flag = np.zeros((n, S), np.uint8)
    stale_grad[::7, 11] = 3.0
    stale_flag[::7, 11] = 1
    ref = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 2.5, out_gradient=stale_grad.copy(),
                                sparse_index=stale_flag.copy())
    assert 0.05 < (ref["distance"] > 0).mean() < 0.95
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    w = torch.tensor([2.5], device=device)
    sc = kp.self_collision
    assert G.pair_bitmap(sc.collision_pairs, S) is not None

    def run(dense):
        out_d, out_g, flags = torch.full((n, 1), -1.0, device=device), t(stale_grad.copy()), t(stale_flag.copy())
        if dense:  # the drop-in signature (dispatches to curobo_hip_self_collision_distance_dense)
            G.self_collision_distance(out_d, out_g, torch.zeros(1, device=device), flags, t(sph), sc.sphere_padding, w, sc.collision_pairs,
                                      torch.zeros(1, device=device), torch.zeros(2, dtype=torch.int16, device=device), 1, 256, B, H, S, P,
                                      False, True)
        else:
            check(load().curobo_hip_self_collision_distance(
                ptr(out_d), ptr(out_g), None, ptr(flags), ptr(t(sph)), ptr(sc.sphere_padding), ptr(w), ptr(sc.collision_pairs), None, None,
                1, 256, B, H, S, P, 0, 1, current_stream(out_d)))
        torch.cuda.synchronize()
        return out_d.cpu().numpy()[:, 0], out_g.cpu().numpy(), flags.cpu().numpy()

    d1, g1, f1 = run(True)
    d0, g0, f0 = run(False)
    # the matrix-core narrow phase (self_collision_tiles_mfma_kernel: two v_mfma_f32_16x16x4_f32 per surviving tile cull, the
    # listed close pairs are evaluated again exactly): every output bit of the two-level kernel
    import os

    os.environ["CUROBO_HIP_SELF_MFMA"] = "1"
    try:
        d2, g2, f2 = run(True)
    finally:
        del os.environ["CUROBO_HIP_SELF_MFMA"]
    assert np.array_equal(d2, d1) and np.array_equal(g2, g1) and np.array_equal(f2, f1)
    assert np.array_equal(f1, ref["sparse_index"]) and np.array_equal(f0, f1), "collision-pair indices must be exact"
    np.testing.assert_allclose(d1, ref["distance"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(g1, ref["gradient"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(d1, d0, atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(g1, g0, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("sweep", [False, True])
def test_voxel_coarse_culling_changes_nothing(sweep, oracle, device):
    """the min-pooled ESDF (curobo_hip_scene.voxel_coarse_min) only skips spheres that provably contribute zero:
    distance and gradient are BIT-identical with and without it (UR10e trajectories through the C3 world)"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c3_voxel_world

    model = load_model("ur10e")
    arrays = c3_voxel_world(64, 0.04)
    b, h = 64, 17
    q0, q1 = sample_q(model, b, seed=3)[:, None], sample_q(model, b, seed=4)[:, None]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    q = (q0 * (1 - tt) + q1 * tt).reshape(b * h, -1)
    sph = oracle.kinematics_forward(q, model.as_dict(), horizon=h)["robot_spheres"].reshape(b, h, -1, 4)
    S = sph.shape[2]
    outs = []
    for coarse in (False, True):
        scene = SceneData.from_arrays(arrays, device, coarse_culling=coarse)
        assert (scene.struct.voxel_coarse_min is not None) == coarse
        dist, grad = torch.full((b, h, S), 3.0, device=device), torch.full((b, h, S, 4), 3.0, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
                                     torch.tensor([0.02], device=device), None, b, h, S, False, 3 if sweep else 0, sweep,
                                     torch.tensor([0.05], device=device) if sweep else None)
        torch.cuda.synchronize()
        outs.append((dist.cpu().numpy(), grad.cpu().numpy()))
    assert 0.02 < (outs[0][0] > 0).mean() < 0.9
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ref = oracle.scene_collision(sph, arrays, 1.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
    assert np.array_equal(outs[1][0] > 0, ref["distance"] > 0)


@pytest.mark.parametrize("sweep,voxel", [(False, False), (True, False), (True, True)])
def test_scene_collision_analytic_primitives(sweep, voxel, oracle, device):
    """sphere / capsule / cylinder records of the cuboid store (closed forms on the device; the reference meshes
    them) next to a cuboid [and an ESDF grid]: HIP kernel vs the oracle, exact hit set"""
    from test_scene_primitives import PRIM_WORLD

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c3_voxel_world

    model = load_model("franka")
    arrays = cuboid_scene_arrays(PRIM_WORLD)
    if voxel:
        arrays = {**arrays, **c3_voxel_world(64, 0.04)}
    b, h = 32, 9
    q0, q1 = sample_q(model, b, seed=11)[:, None], sample_q(model, b, seed=12)[:, None]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    sph = oracle.kinematics_forward((q0 * (1 - tt) + q1 * tt).reshape(b * h, -1) * 0.7, model.as_dict(), horizon=h)["robot_spheres"]
    sph = sph.reshape(b, h, -1, 4)
    S = sph.shape[2]
    ref = oracle.scene_collision(sph, arrays, 3.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
    scene = SceneData.from_arrays(arrays, device)
    assert scene.struct.cuboid_has_primitives == 1
    dist, grad = torch.full((b, h, S), 5.0, device=device), torch.full((b, h, S, 4), 5.0, device=device)
    Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([3.0], device=device),
                                 torch.tensor([0.02], device=device), None, b, h, S, False, 3 if sweep else 0, sweep,
                                 torch.tensor([0.05], device=device))
    torch.cuda.synchronize()
    assert (ref["distance"] > 0).mean() > 0.03
    assert_scene_kernel_parity(oracle, dist.cpu().numpy(), grad.cpu().numpy(), sph, arrays, 3.0, 0.02, f"primitives sweep={sweep} voxel={voxel}",
                               voxel=voxel, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)


def test_mesh_esdf_bake_on_device(oracle, device):
    """mesh -> ESDF on the device (csrc/mesh_bake.hip) vs the NumPy mesh signed distance (same algorithm) and vs the
    closed form of the shape the mesh represents (a rotated box); then the baked grid through the collision kernel"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, bake_esdf, bake_mesh_esdf_device, box_mesh, cuboid_sdf, mesh_sdf

    dims, pose = [0.3, 0.5, 0.2], [0.1, -0.05, 0.4, 0.9238795, 0, 0.3826834, 0]
    v, f = box_mesh(dims)
    lo, hi, vs = [-0.4, -0.5, 0.0], [0.6, 0.4, 0.8], 0.025
    grid = bake_mesh_esdf_device(v, f, pose, lo, hi, vs, device, max_distance=10.0)
    torch.cuda.synchronize()
    host = bake_esdf(mesh_sdf(v, f, pose), lo, hi, vs, max_distance=10.0)
    got = grid["voxel_features"].cpu().numpy().astype(np.float32).reshape(-1)
    want = host["voxel_features"].astype(np.float32).reshape(-1)
    assert np.array_equal(grid["voxel_params"], host["voxel_params"])
    np.testing.assert_allclose(got, want, atol=2e-3)  # fp16 grid, f32 vs f64 arithmetic
    assert np.array_equal(np.sign(got)[np.abs(want) > 2e-3], np.sign(want)[np.abs(want) > 2e-3])
    exact = bake_esdf(cuboid_sdf(dims, pose), lo, hi, vs, max_distance=10.0)["voxel_features"].astype(np.float32).reshape(-1)
    np.testing.assert_allclose(got, exact, atol=2e-3)
    # the device-baked grid as an obstacle: same cost as the host-baked grid
    scene_d, scene_h = SceneData.from_arrays(grid, device), SceneData.from_arrays(host, device)
    rng = np.random.default_rng(0)
    sph = np.concatenate([rng.uniform(lo, hi, size=(4, 3, 50, 3)), rng.uniform(0.02, 0.06, size=(4, 3, 50, 1))], -1).astype(np.float32)
    outs = []
    for scene in (scene_d, scene_h):
        dist, grad = torch.zeros(4, 3, 50, device=device), torch.zeros(4, 3, 50, 4, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
                                     torch.tensor([0.01], device=device), None, 4, 3, 50, False, 0, False, None)
        torch.cuda.synchronize()
        outs.append(dist.cpu().numpy())
    assert (outs[1] > 0).mean() > 0.02
    np.testing.assert_allclose(outs[0], outs[1], atol=3e-3)


def test_empty_batches_are_no_ops(device):
    """zero trajectories / points through the entry points of the path: status OK, nothing launched that reads or writes
    (the reference's wrappers are called with whatever batch the planner has left after its filters -- an empty one included)"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.backends import geometry as G
    from curobo_amd.backends import kinematics as K
    from curobo_amd.backends import trajectory as Tr
    from curobo_amd.scene import SceneData

    model = load_model("franka")
    kp = _kp(model, device)
    d, S, L, T = model.num_dof, model.num_spheres, model.num_links, kp.num_pose_links
    P = model.collision_pairs.shape[0]
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
    env0 = z(0, dt=torch.int32)
    # forward kinematics + spheres, and its VJP
    K.launch_kinematics_forward_spheres(z(0, 1, T, 3), z(0, 1, T, 4), z(0, 1, S, 4), z(0, 1, 4), z(0, 1, L, 3, 4), z(0, d),
                                        kp.fixed_transforms, kp.link_spheres, kp.link_masses_com, kp.joint_map_type, kp.joint_map,
                                        kp.link_map, kp.tool_frame_map, kp.link_sphere_idx_map, kp.joint_offset_map, env0, kp.num_envs,
                                        0, 1, d, S, 32, True, False)
    K.launch_kinematics_backward(z(0, d), z(0, T, 3), z(0, T, 4), z(0, S, 4), z(0, 4), z(0, 4), z(0, T, 3), z(0, L, 3, 4),
                                 kp.link_spheres, kp.link_masses_com, kp.link_map, kp.joint_map, kp.joint_map_type, kp.tool_frame_map,
                                 kp.link_sphere_idx_map, kp.link_chain_data, kp.link_chain_offsets, kp.joint_links_data,
                                 kp.joint_links_offsets, kp.joint_affects_endeffector, kp.joint_offset_map, env0, 1, 0, 1, d, S, False,
                                 False)
    # self collision, scene collision (cuboids)
    G.self_collision_distance(z(0, 1), z(0, S, 4), z(0, P), z(0, S, dt=torch.uint8), z(0, S, 4), kp.self_collision.sphere_padding,
                              torch.tensor([2.5], device=device), kp.self_collision.collision_pairs, z(1), z(2, dt=torch.int16), 1, 256,
                              0, 1, S, P, True, True)
    scene = SceneData.from_arrays(_scene_arrays(), device)
    Cn.sphere_obstacle_collision(z(0, 3, S), z(0, 3, S, 4), z(0, 3, S, 4), scene.struct, torch.tensor([3.0], device=device),
                                 torch.tensor([0.03], device=device), env0, 0, 3, S, False, 3, True, torch.tensor([0.05], device=device))
    # B-spline forward / VJP
    nk, degree, interp = 12, 3, 2
    ph = (nk + degree + 1) * interp + 1
    st = [z(1, d) for _ in range(8)]
    Tr.launch_bspline_interpolation_forward_kernel(z(0, ph, d), z(0, ph, d), z(0, ph, d), z(0, ph, d), z(0), z(0, nk, d), *st,
                                                   env0, env0, torch.tensor([0.05], device=device), z(1, dt=torch.uint8), 0, ph, d, nk,
                                                   degree)
    Tr.launch_bspline_interpolation_backward_kernel(z(0, nk, d), z(0, ph, d), z(0, ph, d), z(0, ph, d), z(0, ph, d),
                                                    torch.tensor([0.05], device=device), env0, z(1, dt=torch.uint8), 0, ph, d, nk, degree,
                                                    False)
    # optimiser side: L-BFGS step, line search, per-trajectory cost sum
    from curobo_amd.backends import optimization as Op

    v, m, nls = 84, 5, 4
    Op.launch_lbfgs_step(z(0, v), z(m, 0), z(m, 0, v), z(m, 0, v), z(0, v), z(0, v), z(0, v), z(0, v), 0.01, 0, m, v, True, True)
    i16 = lambda *s: z(*s, dt=torch.int16)  # noqa: E731
    Op.launch_line_search(z(0), z(0, v), i16(0), i16(0), z(0, dt=torch.uint8), 5, 1e-4, 1e-3, z(0), z(0, v), z(0, v),
                          z(0, nls, dt=torch.int32), z(0), z(0, v), z(0, v), z(0, nls, dt=torch.int32), z(0, nls, 1), z(0, nls, v),
                          z(0, nls, v), z(0, 1, v), torch.tensor([0.0, 0.1, 0.5, 1.0], device=device), 1e-5, 0.9, False, True, nls, v, 0)
    Cn.trajectory_cost_sum(z(0), z(0, 3), z(0, 3, S), 0, 3, S)
    torch.cuda.synchronize()
