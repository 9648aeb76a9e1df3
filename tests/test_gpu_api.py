"""Drop-in layer on the GPU: autograd wrappers (reference cuda_ops / Warp autograd contract) and
the public Kinematics / RobotCollisionChecker API give the same numbers as the explicit rollout."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q

pytestmark = pytest.mark.gpu


def _cfg(device, name="franka"):
    from curobo_amd.kinematics import KinematicsCfg

    return KinematicsCfg.from_packaged(name, device)


def test_kinematics_api_matches_oracle(oracle, device):
    from curobo_amd.kinematics import Kinematics

    cfg = _cfg(device)
    kin = Kinematics(cfg, compute_jacobian=True, compute_spheres=True, compute_com=True)
    q = sample_q(cfg.model, 12, seed=1).reshape(4, 3, 7)
    st = kin.compute_kinematics(torch.as_tensor(q, device=device))
    ref = oracle.kinematics_forward(q.reshape(12, 7), cfg.model.as_dict(), horizon=3, compute_jacobian=True,
                                    compute_com=True)
    np.testing.assert_allclose(st.tool_poses.position.cpu().numpy().reshape(12, 1, 3), ref["link_pos"], atol=1e-5)
    np.testing.assert_allclose(st.tool_poses.quaternion.cpu().numpy().reshape(12, 1, 4), ref["link_quat"], atol=1e-5)
    np.testing.assert_allclose(st.robot_spheres.cpu().numpy().reshape(12, 65, 4), ref["robot_spheres"], atol=1e-5)
    np.testing.assert_allclose(st.tool_jacobians.cpu().numpy().reshape(12, 1, 6, 7), ref["jacobian"], atol=1e-5)
    assert st.tool_poses.tool_frames == ["panda_hand"]
    # get_link_poses / update_batch_size (reference kinematics.py:75-100, 278-311)
    kin.update_batch_size(12, 1)
    lp = kin.get_link_poses(torch.as_tensor(q.reshape(12, 7), device=device), ["panda_hand"])
    np.testing.assert_allclose(lp.position.cpu().numpy().reshape(12, 1, 3), ref["link_pos"], atol=1e-5)
    np.testing.assert_allclose(lp.quaternion.cpu().numpy().reshape(12, 1, 4), ref["link_quat"], atol=1e-5)
    with pytest.raises(ValueError, match="not tool frames"):
        kin.get_link_poses(torch.zeros(1, 7, device=device), ["panda_link3"])


def test_autograd_through_kinematics_matches_oracle_vjp(oracle, device):
    from curobo_amd.kinematics import Kinematics

    cfg = _cfg(device)
    kin = Kinematics(cfg, compute_spheres=True)
    rng = np.random.default_rng(0)
    qn = sample_q(cfg.model, 10, seed=2).reshape(5, 2, 7)
    q = torch.as_tensor(qn, device=device).requires_grad_(True)
    st = kin.compute_kinematics(q)
    gs = rng.normal(size=(5, 2, 65, 4)).astype(np.float32)
    gp = rng.normal(size=(5, 2, 1, 3)).astype(np.float32)
    loss = (st.robot_spheres * torch.as_tensor(gs, device=device)).sum() + \
        (st.tool_poses.position * torch.as_tensor(gp, device=device)).sum()
    loss.backward()
    fwd = oracle.kinematics_forward(qn.reshape(10, 7), cfg.model.as_dict(), horizon=2)
    ref = oracle.kinematics_backward(cfg.model.as_dict(), fwd["cumul_mat"], gs.reshape(10, 65, 4), gp.reshape(10, 1, 3),
                                     horizon=2)
    np.testing.assert_allclose(q.grad.cpu().numpy().reshape(10, 7), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_collision_checker_autograd_equals_explicit_rollout(device):
    """cost.backward() through the autograd wrappers == the explicit VJP path of CollisionRollout"""
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    cfg = _cfg(device)
    model = cfg.model
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    rcfg = CollisionRolloutCfg(use_sweep=False, use_speed_metric=False, use_fused=False)
    knots = seed_knots(model, 8, rcfg.n_knots, seed=5)
    ro = CollisionRollout(cfg.kinematics_config, scene, 8, rcfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    cost, _ = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(8, -1))
    q_traj = ro.position.clone()
    grad_q_explicit = ro.grad_q.clone()
    chk = RobotCollisionChecker(cfg, scene, activation_distance=rcfg.activation_distance,
                                scene_weight=rcfg.scene_collision_weight, self_weight=rcfg.self_collision_weight)
    q = q_traj.clone().requires_grad_(True)
    d_world, d_self = chk.get_scene_self_collision_distance_from_joints(q)
    total = d_world.sum((1, 2)) + d_self.sum((1, 2))
    torch.testing.assert_close(total, cost, rtol=1e-5, atol=1e-3)
    total.sum().backward()
    torch.testing.assert_close(q.grad, grad_q_explicit, rtol=1e-4, atol=1e-5 * float(grad_q_explicit.abs().max()))
    free = chk.validate(q_traj)
    assert free.shape == (8, rcfg.padded_horizon) and free.any() and (~free).any()


def test_bspline_and_lbfgs_autograd_wrappers(oracle, device):
    from curobo_amd.hip_ops import BSplineIdxKernel, LBFGScu

    rng = np.random.default_rng(1)
    b, nk, dof, deg, interp = 6, 12, 7, 3, 2
    ph = (nk + deg + 1) * interp + 1
    t = lambda a, dt=torch.float32: torch.as_tensor(a, device=device).to(dt)  # noqa: E731
    u = t(rng.normal(size=(b, nk, dof)).astype(np.float32)).requires_grad_(True)
    z = lambda: torch.zeros(1, dof, device=device)  # noqa: E731
    outs = [torch.zeros(b, ph, dof, device=device) for _ in range(4)]
    idx = torch.zeros(b, dtype=torch.int32, device=device)
    p, v, a_, j = BSplineIdxKernel.apply(u, z(), z(), z(), z(), z(), z(), z(), z(), idx, idx, *outs,
                                         torch.zeros(b, device=device), t([0.05]), torch.zeros(1, dtype=torch.uint8, device=device),
                                         torch.zeros(b, nk, dof, device=device), deg)
    gp = rng.normal(size=(b, ph, dof)).astype(np.float32)
    gv = rng.normal(size=(b, ph, dof)).astype(np.float32) * 0.1
    (p * t(gp)).sum().add((v * t(gv)).sum()).backward()
    zz = np.zeros_like(gp)
    ref = oracle.bspline_backward(gp, gv, zz, zz, np.array([0.05], np.float32), np.zeros(b, np.int32),
                                  np.zeros(1, np.uint8), nk, deg)
    np.testing.assert_allclose(u.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    # LBFGScu with the reference's QuasiNewtonBuffers shapes (m,B,V,1)/(m,B,1,1)/(B,V,1)
    m, V = 5, 20
    mk = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
    st = dict(step=np.zeros((b, V), np.float32), rho=np.abs(mk(m, b)) * 0.1, y=mk(m, b, V), s=mk(m, b, V),
              q=mk(b, V), g=mk(b, V), x0=mk(b, V), g0=mk(b, V))
    dv = {k: t(a.copy()) for k, a in st.items()}
    out = LBFGScu.apply(dv["step"], dv["rho"].view(m, b, 1, 1), dv["y"].view(m, b, V, 1), dv["s"].view(m, b, V, 1),
                        dv["q"], dv["g"].view(b, 1, V), dv["x0"].view(b, V, 1), dv["g0"].view(b, V, 1), 0.01, True, True)
    oracle.lbfgs_step(st["step"], st["rho"], st["y"], st["s"], st["q"], st["g"], st["x0"], st["g0"], 0.01, True)
    np.testing.assert_allclose(out.cpu().numpy(), st["step"], rtol=2e-4, atol=2e-5 * np.abs(st["step"]).max())


def test_tensor_checks_reject_bad_inputs(device):
    from curobo_amd.hip_ops.tensor_checks import check_float32_tensors

    good = torch.zeros(4, 4, device=device)
    check_float32_tensors(device, good=good)
    with pytest.raises(ValueError, match="not contiguous"):
        check_float32_tensors(device, bad=good.t())
    with pytest.raises(ValueError, match="dtype"):
        check_float32_tensors(device, bad=good.double())
    with pytest.raises(ValueError, match="is on"):
        check_float32_tensors(device, bad=torch.zeros(2))


def test_c3_ur10e_voxel_world_collision_checking_path(oracle, device):
    """BASELINE config 3: UR10e + one nvblox-style ESDF voxel grid (128^3 at 0.02 m = 2.56 m cube,
    fp16), 512 seeds through the collision_checking SDF path
    (RobotCollisionChecker.get_scene_self_collision_distance_from_joints) vs the oracle."""
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.scene import SceneData, voxel_grid_from_sdf

    def sdf(p):  # union of a box and a sphere, like the reference's analytic voxel suites
        qb = np.abs(p - np.array([0.5, 0.0, 0.35])) - np.array([0.2, 0.4, 0.35])
        box = np.linalg.norm(np.maximum(qb, 0), axis=-1) + np.minimum(qb.max(-1), 0)
        sp = np.linalg.norm(p - np.array([-0.35, 0.45, 0.6]), axis=-1) - 0.25
        return np.minimum(box, sp)

    cfg = KinematicsCfg.from_packaged("ur10e", device=device)
    model = cfg.model
    arrays = voxel_grid_from_sdf(sdf, (128, 128, 128), 0.02, pose7=(0.0, 0.0, 0.6, 1, 0, 0, 0), max_distance=10.0)
    assert arrays["voxel_features"].shape[-1] == 128 ** 3 and arrays["voxel_features"].dtype == np.float16
    scene = SceneData.from_arrays(arrays, device)
    seeds, horizon = 512, 4
    q = sample_q(model, seeds * horizon, seed=33).astype(np.float32).reshape(seeds, horizon, -1)
    chk = RobotCollisionChecker(cfg, scene, activation_distance=0.02, scene_weight=1.0, self_weight=1.0)
    d_world, d_self = chk.get_scene_self_collision_distance_from_joints(torch.as_tensor(q, device=device))
    torch.cuda.synchronize()
    assert d_world.shape == (seeds, horizon, model.num_spheres) and d_self.shape == (seeds, horizon, 1)
    fk = oracle.kinematics_forward(q.reshape(-1, q.shape[-1]), model.as_dict(), horizon=horizon)
    sph = fk["robot_spheres"].reshape(seeds, horizon, -1, 4)
    ref_w = oracle.scene_collision(sph, arrays, 1.0, 0.02)
    ref_s = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)
    assert (ref_w["distance"] > 0).mean() > 0.02, "the synthetic world must produce hits"
    # (FK on the device vs FK of the oracle: the spheres differ by ~1e-6 m, the cost of the quadratic zone by that much;
    # on IDENTICAL spheres the kernel is held to 1e-5 + 5e-6 m in test_gpu_kernels.py::test_scene_collision_voxels)
    err = np.abs(d_world.cpu().numpy() - ref_w["distance"]) / (1e-5 * np.abs(ref_w["distance"]) + 1e-5)
    print(f"\n[c3 checker] worst scene cost error {float(err.max()):.3f} of (1e-5 rel + 1e-5 m); FK-to-FK sphere difference included")
    np.testing.assert_allclose(d_world.cpu().numpy(), ref_w["distance"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d_self.cpu().numpy().reshape(-1), ref_s["distance"], rtol=1e-5, atol=1e-6)
    # bit-exact collision-hit indices (north_star): which spheres are in collision
    assert np.array_equal(d_world.cpu().numpy() > 0, ref_w["distance"] > 0)


def test_cspace_l2_distance_autograd(device):
    """reference L2DistFunction (cost/wp_torch_cspace_dist.py): cost / gradient vs the closed form,
    zero-weight entries untouched, gradient scaling through use_grad_input"""
    from curobo_amd.hip_ops import L2DistFunction

    torch.manual_seed(0)
    B, H, D = 5, 4, 7
    pos = torch.randn(B, H, D, device=device, requires_grad=True)
    target = torch.randn(3, D, device=device)
    tidx = torch.tensor([2, 0, 1, 1, 0], dtype=torch.int32, device=device)
    w = torch.tensor([3.0], device=device)
    term = torch.rand(D, device=device) + 0.5
    nonterm = torch.rand(D, device=device)
    nonterm[2] = 0.0  # untouched entries
    out_c = torch.full((B, H, D), -7.0, device=device)
    out_g = torch.full((B, H, D), -7.0, device=device)
    cost = L2DistFunction.apply(pos, target, tidx, w, term, nonterm, out_c, out_g, True)
    r = nonterm.view(1, 1, D).expand(B, H, D).clone()
    r[:, -1] = term
    err = pos.detach() - target[tidx.long()].view(B, 1, D)
    ref = 3.0 * r * err * err
    mask = (r != 0)
    torch.testing.assert_close(out_c[mask], ref[mask], rtol=1e-6, atol=1e-6)
    assert bool((out_c[~mask] == -7.0).all()) and bool((out_g[~mask] == -7.0).all())
    out_c[~mask] = 0.0
    out_g[~mask] = 0.0
    scale = torch.rand(B, H, device=device)
    cost = L2DistFunction.apply(pos, target, tidx, w, term, nonterm, out_c, out_g, True)
    (cost * scale).sum().backward()
    torch.testing.assert_close(pos.grad, 6.0 * r * err * scale.unsqueeze(-1), rtol=1e-5, atol=1e-6)


def test_collision_checker_validate_sample_and_bounds(oracle, device):
    """RobotCollisionChecker.validate / validate_trajectory / get_bound / sample / sample_trajectory
    (reference collision_robot_scene.py:286-417) against the oracle's distances."""
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world

    model = load_model("franka")
    md = model.as_dict()
    arrays = cuboid_scene_arrays(c2_world())
    chk = RobotCollisionChecker(KinematicsCfg.from_packaged("franka", device=device), SceneData.from_arrays(arrays, device))
    B, H = 6, 5
    q = sample_q(model, B * H, seed=12, scale=1.08).reshape(B, H, 7)  # some rows beyond the limits
    tq = torch.as_tensor(q, device=device)
    ok = chk.validate(tq).cpu().numpy()
    assert ok.shape == (B, H) and np.array_equal(ok, chk.validate_trajectory(tq).cpu().numpy())
    fk = oracle.kinematics_forward(q.reshape(-1, 7), md)
    sph = fk["robot_spheres"].reshape(B, H, -1, 4)
    d_self = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"].reshape(B, H)
    d_world = oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].reshape(B, H, -1).sum(-1)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    inside = ((q >= lo) & (q <= hi)).all(-1)
    want = (d_self == 0) & (d_world == 0) & inside
    assert np.array_equal(ok, want) and want.any() and (~want).any()
    bound = chk.get_bound(tq).cpu().numpy()
    viol = np.maximum(q - hi, 0) + np.maximum(lo - q, 0)
    np.testing.assert_allclose(bound, 0.5 * viol * viol, rtol=1e-5, atol=1e-7)
    # sampling: Halton points in the limits; with mask_valid every returned configuration validates
    qs = chk.sample(32, mask_valid=True)
    assert 0 < qs.shape[0] <= 32 and qs.shape[1] == 7 and bool(chk.validate(qs.unsqueeze(1)).all())
    raw = chk.sample(64, mask_valid=False)
    assert raw.shape == (64, 7) and bool(((raw >= torch.as_tensor(lo, device=device)) & (raw <= torch.as_tensor(hi, device=device))).all())
    traj = chk.sample_trajectory(3, 4, mask_valid=True)
    assert traj.shape == (3, 4, 7) and bool(chk.validate_trajectory(traj).all())


def test_cost_modules_match_oracle_and_backpropagate(oracle, device):
    """SelfCollisionCost / SceneCollisionCost (reference cost/cost_self_collision.py, cost_scene_collision.py):
    forward values vs the oracle, sum / max / binary variants, gradients through the kinematics."""
    from curobo_amd.cost import SceneCollisionCost, SceneCollisionCostCfg, SelfCollisionCost, SelfCollisionCostCfg
    from curobo_amd.kinematics import Kinematics, KinematicsCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world

    model = load_model("franka")
    md = model.as_dict()
    cfg = KinematicsCfg.from_packaged("franka", device=device)
    kin = Kinematics(cfg, compute_spheres=True)
    arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(arrays, device)
    B, H = 5, 4
    q = sample_q(model, B * H, seed=21).reshape(B, H, 7)
    tq = torch.as_tensor(q, device=device).requires_grad_(True)
    sph = kin.compute_kinematics(tq).robot_spheres
    S = sph.shape[2]
    self_cost = SelfCollisionCost(SelfCollisionCostCfg(cfg.kinematics_config.self_collision, weight=3.0), device)
    with pytest.raises(ValueError):
        self_cost.forward(sph)  # buffers not set up
    self_cost.setup_batch_tensors(B, H)
    scene_cost = SceneCollisionCost(SceneCollisionCostCfg(scene, S, weight=2.0, activation_distance=0.02), device)
    scene_cost.setup_batch_tensors(B, H)
    cs, cw = self_cost(sph), scene_cost(sph)
    assert cs.shape == (B, H) and cw.shape == (B, H)
    fk = oracle.kinematics_forward(q.reshape(-1, 7), md)
    ref_s = oracle.self_collision(fk["robot_spheres"].reshape(B, H, S, 4), model.sphere_padding, model.collision_pairs, 3.0)
    ref_w = oracle.scene_collision(fk["robot_spheres"].reshape(B, H, S, 4), arrays, 2.0, 0.02)
    np.testing.assert_allclose(cs.detach().cpu().numpy(), ref_s["distance"].reshape(B, H), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cw.detach().cpu().numpy(), ref_w["distance"].reshape(B, H, S).sum(-1), rtol=1e-5, atol=1e-5)
    (cs.sum() + cw.sum()).backward()
    gs = ref_s["gradient"].reshape(B * H, S, 4) + ref_w["gradient"].reshape(B * H, S, 4) * np.array([1, 1, 1, 0], np.float32)
    ref_g = oracle.kinematics_backward(md, fk["cumul_mat"], gs)
    np.testing.assert_allclose(tq.grad.cpu().numpy().reshape(-1, 7), ref_g, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref_g).max()))
    assert (ref_s["distance"] > 0).any() and (ref_w["distance"] > 0).any()
    # variants: worst sphere instead of the sum, binary flags
    mx = SceneCollisionCost(SceneCollisionCostCfg(scene, S, weight=2.0, activation_distance=0.02, sum_distance=False), device)
    mx.setup_batch_tensors(B, H)
    np.testing.assert_allclose(mx(sph.detach()).cpu().numpy(), ref_w["distance"].reshape(B, H, S).max(-1), rtol=1e-5, atol=1e-5)
    bn = SelfCollisionCost(SelfCollisionCostCfg(cfg.kinematics_config.self_collision, weight=3.0, convert_to_binary=True), device)
    bn.setup_batch_tensors(B, H)
    b = bn(sph.detach()).cpu().numpy()
    want = np.minimum(ref_s["distance"].reshape(B, H), 1.0)
    np.testing.assert_allclose(b, np.where(want > 0, want + 1.0, want), rtol=1e-5, atol=1e-5)


def test_tool_pose_cost_module_autograd(oracle, device):
    """ToolPoseCost / ToolPoseDistance (reference cost_tool_pose.py, wp_tool_pose.py:696-914): values vs the
    oracle, gradient to the joint angles through the kinematics vs the oracle's FK VJP of the pose gradients."""
    from curobo_amd.cost import ToolPoseCost, ToolPoseCostCfg
    from curobo_amd.kinematics import Kinematics, KinematicsCfg

    model = load_model("franka")
    md = model.as_dict()
    kin = Kinematics(KinematicsCfg.from_packaged("franka", device=device), compute_spheres=False)
    B, H, T, G = 6, 3, 1, 2
    q = sample_q(model, B * H, seed=31).reshape(B, H, 7)
    goals = oracle.kinematics_forward(sample_q(model, 4 * G, seed=32), md, compute_spheres=False)
    gp = goals["link_pos"].reshape(4, T, G, 3)
    gq = goals["link_quat"].reshape(4, T, G, 4)
    idx = np.array([0, 3, 1, 2, 2, 0], np.int32)
    cost = ToolPoseCost(ToolPoseCostCfg(num_links=T, weight=[20.0, 5.0], non_terminal_pose_axes_weight_factor=[0.5] * 6), device)
    cost.setup_batch_tensors(B, H)
    tq = torch.as_tensor(q, device=device).requires_grad_(True)
    st = kin.compute_kinematics(tq)
    pos, quat = st.tool_poses.position.view(B, H, T, 3), st.tool_poses.quaternion.view(B, H, T, 4)
    c, lin, ang, gidx = cost(pos, quat, torch.as_tensor(gp, device=device), torch.as_tensor(gq, device=device),
                             torch.as_tensor(idx, device=device))
    fk = oracle.kinematics_forward(q.reshape(-1, 7), md, compute_spheres=False)
    ref = oracle.tool_pose_distance(fk["link_pos"].reshape(B, H, T, 3), fk["link_quat"].reshape(B, H, T, 4), gp, gq, idx,
                                    np.array([20.0, 5.0], np.float32), np.ones(6 * T, np.float32), np.full(6 * T, 0.5, np.float32),
                                    np.zeros(2 * T, np.float32), np.zeros(2 * T, np.float32), np.zeros(T, np.uint8), 0)
    np.testing.assert_allclose(c.detach().cpu().numpy(), ref["distance"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lin.cpu().numpy(), ref["position_distance"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ang.cpu().numpy(), ref["rotation_distance"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(gidx.cpu().numpy(), ref["goalset_idx"]) and set(np.unique(ref["goalset_idx"])) == {0, 1}
    c.sum().backward()
    ref_g = oracle.kinematics_backward(md, fk["cumul_mat"], None, ref["position_gradient"].reshape(B * H, T, 3),
                                       ref["rotation_gradient"].reshape(B * H, T, 4))
    np.testing.assert_allclose(tq.grad.cpu().numpy().reshape(-1, 7), ref_g, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref_g).max()))


def test_state_cspace_function_autograd(oracle, device):
    """StateCSpaceFunction (reference wp_cspace_state.py:288-680): cost vs the oracle and gradients to all
    five state streams, scaled by the incoming gradient (use_grad_input)."""
    from curobo_amd.hip_ops import StateCSpaceFunction

    rng = np.random.default_rng(5)
    B, H, D = 4, 6, 7
    model = load_model("franka")
    lim_p = np.asarray(model.joint_limits_position, np.float32)
    f = lambda s: rng.normal(size=(B, H, D)).astype(np.float32) * s  # noqa: E731
    pos, vel, acc, jerk, tau = f(2.0), f(2.0), f(8.0), f(200.0), f(30.0)
    lim = {"position": lim_p, "velocity": np.stack([-np.ones(D), np.ones(D)]).astype(np.float32) * 2.0,
           "acceleration": np.stack([-np.ones(D), np.ones(D)]).astype(np.float32) * 10.0,
           "jerk": np.stack([-np.ones(D), np.ones(D)]).astype(np.float32) * 300.0,
           "effort": np.stack([-np.ones(D), np.ones(D)]).astype(np.float32) * 40.0}
    weight = np.array([100.0, 10.0, 5.0, 1.0, 3.0], np.float32)
    eta = np.array([0.05, 0.0, 0.0, 0.0, 0.1], np.float32)
    sql2 = np.array([0.1, 0.05, 0.01, 0.02, 0.0], np.float32)
    dt = np.full(B, 0.05, np.float32)
    ref = oracle.cspace_state_cost(pos, vel, acc, jerk, dt, lim, weight, eta, sql2, effort=tau)
    t = lambda a, dt_=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dt_)  # noqa: E731
    z = lambda: torch.zeros(B, H, D, device=device)  # noqa: E731
    ins = [t(x).requires_grad_(True) for x in (pos, vel, acc, jerk, tau)]
    cost = StateCSpaceFunction.apply(
        *ins, t(dt), torch.zeros(1, D, device=device), torch.zeros(B, dtype=torch.int32, device=device), t(lim["position"]),
        t(lim["velocity"]), t(lim["acceleration"]), t(lim["jerk"]), t(lim["effort"]), t(weight), t(eta), t(sql2),
        torch.zeros(1, device=device), torch.ones(1, device=device), torch.ones(D, device=device), z(), z(), z(), z(), z(), z(),
        False, False, True)
    np.testing.assert_allclose(cost.detach().cpu().numpy(), ref["cost"], rtol=1e-5, atol=1e-5)
    scale = torch.rand(B, H, D, device=device) + 0.5
    (cost * scale).sum().backward()
    for x, k in zip(ins, ("grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")):
        np.testing.assert_allclose(x.grad.cpu().numpy(), ref[k] * scale.cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref[k]).max()), err_msg=k)
        assert np.abs(ref[k]).max() > 0


@pytest.mark.parametrize("robot", ["franka", "ur10e", "unitree_g1"])
def test_jacobian_output_is_differentiable(robot, device):
    """dJ/dq: gradient THROUGH the geometric-Jacobian output (reference JAC_GRAD, kinematics_backward_kernel.cuh:27-157,
    kinematics_jacobian_backward_helper.cuh) vs central finite differences of the Jacobian itself (the reference's
    tests/_src/robot/kinematics/test_jacobian_gradcheck.py style), serial arm, and a humanoid tree with four tool frames"""
    from curobo_amd.kinematics import Kinematics, KinematicsCfg

    cfg = KinematicsCfg.from_packaged(robot, device=device)
    kin = Kinematics(cfg, compute_jacobian=True, compute_spheres=False)
    D, T = cfg.kinematics_config.num_dof, cfg.kinematics_config.num_pose_links
    rng = np.random.default_rng(3)
    qn = sample_q(cfg.model, 6, seed=9, scale=0.7).reshape(3, 2, D)
    w = torch.as_tensor(rng.normal(size=(3, 2, T, 6, D)).astype(np.float32), device=device)
    q = torch.as_tensor(qn, device=device).requires_grad_(True)
    st = kin.compute_kinematics(q)
    (st.tool_jacobians * w).sum().backward()
    got = q.grad.cpu().numpy().astype(np.float64)
    assert np.abs(got).max() > 1e-3
    eps = 1e-3
    fd = np.zeros_like(got)
    with torch.no_grad():
        for d in range(D):
            dq = torch.zeros(3, 2, D, device=device)
            dq[..., d] = eps
            jp = kin.compute_kinematics(torch.as_tensor(qn, device=device) + dq).tool_jacobians.double().clone()
            jm = kin.compute_kinematics(torch.as_tensor(qn, device=device) - dq).tool_jacobians.double().clone()
            fd[..., d] = (((jp - jm) / (2 * eps)) * w.double()).sum(dim=(2, 3, 4)).cpu().numpy()
    np.testing.assert_allclose(got, fd, rtol=2e-2, atol=2e-2 * np.abs(fd).max())
    # and together with a pose gradient: the two VJPs add up
    q2 = torch.as_tensor(qn, device=device).requires_grad_(True)
    st2 = kin.compute_kinematics(q2)
    wp = torch.as_tensor(rng.normal(size=(3, 2, T, 3)).astype(np.float32), device=device)
    ((st2.tool_jacobians * w).sum() + (st2.tool_poses.position * wp).sum()).backward()
    q3 = torch.as_tensor(qn, device=device).requires_grad_(True)
    (kin.compute_kinematics(q3).tool_poses.position * wp).sum().backward()
    torch.testing.assert_close(q2.grad, q.grad + q3.grad, rtol=1e-4, atol=1e-4 * float(q.grad.abs().max()))


def test_collision_checker_sphere_level_api(oracle, device):
    """The sphere-level entry points of the reference's RobotSceneCollision (collision_robot_scene.py:70-245, :418-495), one by
    one against the oracle: get_kinematics, get_collision_distance / _constraint / _vector, get_self_collision(_distance),
    pose_distance, get_point_robot_distance, clear_scene_cache."""
    from sweep_allowance import device_frame_arithmetic

    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.types import Pose
    from curobo_amd.workloads import c2_world

    model = load_model("franka")
    md = model.as_dict()
    arrays = cuboid_scene_arrays(c2_world())
    eta = 0.02
    chk = RobotCollisionChecker(KinematicsCfg.from_packaged("franka", device=device), SceneData.from_arrays(arrays, device),
                                activation_distance=eta)
    B, H = 12, 4
    q = sample_q(model, B * H, seed=4, scale=0.9).reshape(B, H, 7)
    tq = torch.as_tensor(q, device=device)
    # ---- get_kinematics
    state = chk.get_kinematics(tq)
    fk = oracle.kinematics_forward(q.reshape(-1, 7), md, horizon=H)
    sph = fk["robot_spheres"].reshape(B, H, -1, 4)
    np.testing.assert_allclose(state.robot_spheres.cpu().numpy(), sph, atol=1e-5)
    with pytest.raises(ValueError, match="batch, horizon, dof"):
        chk.get_kinematics(tq[:, 0])
    # the collision entry points are checked on the ORACLE's spheres (identical inputs)
    ts = torch.as_tensor(sph, device=device)
    # ---- scene: cost (activation shell), constraint (eta = 0), vector
    with device_frame_arithmetic(oracle):
        ref = oracle.scene_collision(sph, arrays, 1.0, eta)
        ref0 = oracle.scene_collision(sph, arrays, 1.0, 0.0)
    d = chk.get_collision_distance(ts).detach().cpu().numpy()
    assert d.shape == (B, H, model.num_spheres) and (ref["distance"] > 0).sum() > 50
    assert np.array_equal(d > 0, ref["distance"] > 0)
    np.testing.assert_allclose(d, ref["distance"], rtol=1e-5, atol=1e-6)
    d_state = chk.get_collision_distance(state).detach().cpu().numpy()  # a kinematics state is accepted as well
    np.testing.assert_allclose(d_state, d, rtol=1e-4, atol=1e-5)  # (device FK vs oracle FK: 1e-6 m apart)
    c = chk.get_collision_constraint(ts).detach().cpu().numpy()
    assert np.array_equal(c > 0, ref0["distance"] > 0) and (c > 0).sum() < (d > 0).sum()
    np.testing.assert_allclose(c, ref0["distance"], rtol=1e-5, atol=1e-6)
    dv, vec = chk.get_collision_vector(ts)
    np.testing.assert_allclose(dv.cpu().numpy(), ref["distance"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vec.cpu().numpy(), ref["gradient"], rtol=1e-3, atol=2e-4)
    # differentiable in the spheres: the gradient of the summed cost is the vector
    ts2 = ts.clone().requires_grad_(True)
    chk.get_collision_distance(ts2).sum().backward()
    np.testing.assert_allclose(ts2.grad.cpu().numpy(), ref["gradient"], rtol=1e-3, atol=2e-4)
    # ---- self collision
    # (configurations folded into themselves: joint angles scaled past the limits, FK does not mind)
    sph_s = oracle.kinematics_forward(q.reshape(-1, 7) * 1.7, md, horizon=H)["robot_spheres"].reshape(B, H, -1, 4)
    rs = oracle.self_collision(sph_s, model.sphere_padding, model.collision_pairs, 1.0)
    ts_s = torch.as_tensor(sph_s, device=device)
    ds = chk.get_self_collision(ts_s).detach().cpu().numpy()
    assert ds.shape == (B, H, 1) and (rs["distance"] > 0).sum() >= 3
    np.testing.assert_allclose(ds.reshape(-1), rs["distance"].reshape(-1), rtol=1e-5, atol=1e-7)
    assert np.array_equal(chk.get_self_collision_distance(ts_s).detach().cpu().numpy(), ds)
    # ---- pose distance: position error + geodesic rotation error
    rng = np.random.default_rng(1)
    p0, p1 = rng.normal(size=(B, 3)).astype(np.float32), rng.normal(size=(B, 3)).astype(np.float32)
    q0, q1 = rng.normal(size=(B, 4)).astype(np.float32), rng.normal(size=(B, 4)).astype(np.float32)
    q0 /= np.linalg.norm(q0, axis=-1, keepdims=True)
    q1 /= np.linalg.norm(q1, axis=-1, keepdims=True)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    pdist = chk.pose_distance(Pose(t(p0), t(q0)), Pose(t(p1), t(q1)), resize=True).cpu().numpy()
    want = oracle.tool_pose_distance(p1.reshape(B, 1, 1, 3), q1.reshape(B, 1, 1, 4), p0.reshape(B, 1, 1, 3), q0.reshape(B, 1, 1, 4),
                                     np.arange(B, dtype=np.int32), np.ones(2, np.float32), np.ones(6, np.float32), np.ones(6, np.float32),
                                     np.zeros(2, np.float32), np.zeros(2, np.float32), np.zeros(1, np.uint8))
    np.testing.assert_allclose(pdist, (want["position_distance"] + want["rotation_distance"]).reshape(B), rtol=1e-4, atol=1e-5)
    same = chk.pose_distance(Pose(t(p0), t(q0)), Pose(t(p0), t(q0)), resize=True).cpu().numpy()
    assert np.abs(same).max() < 1e-3
    # ---- points against the robot: depth inside the union of its spheres
    q1c = tq[:1, 0]
    pts = rng.uniform(-0.6, 0.8, size=(500, 3)).astype(np.float32)
    s1 = sph[0, 0]
    live = s1[:, 3] >= 0
    pts[:40] = s1[live][:40, :3] + rng.normal(size=(40, 3)).astype(np.float32) * 0.01  # some points inside the robot
    got = chk.get_point_robot_distance(t(pts), q1c).cpu().numpy()
    wantp = (s1[live, 3][None] - np.linalg.norm(pts[:, None] - s1[live, :3][None], axis=-1)).max(-1)
    np.testing.assert_allclose(got, wantp, atol=1e-5)
    assert (got > 0).any() and (got < 0).any()
    gotb = chk.get_point_robot_distance(t(pts).view(2, 250, 3), q1c).cpu().numpy()
    np.testing.assert_allclose(gotb.reshape(-1), wantp, atol=1e-5)
    with pytest.raises(ValueError, match=r"\[b, dof\]"):
        chk.get_point_robot_distance(t(pts), q1c[0])
    # ---- clear_scene_cache: an empty world of the same capacity
    assert chk.tool_frames == chk.kinematics.tool_frames
    chk.clear_scene_cache()
    assert float(chk.get_collision_distance(ts).abs().sum()) == 0.0


def test_front_ends_sample_collision_free_configurations(oracle, device):
    """``sample_configs`` (reference solver_core.py:447-480) on the trajectory optimiser, the planner and the controller: inside the
    limits, free of self and scene collision by the oracle"""
    from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg
    from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo_amd.scene.types import Cuboid
    from curobo_amd.scene.types import SceneCfg as Scene

    scene = Scene(cuboid=[Cuboid(name="table", dims=[0.6, 1.0, 0.1], pose=[0.7, 0.0, 0.2, 1, 0, 0, 0])])
    planner = MotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=scene))
    mpc = ModelPredictiveControl(ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=scene))
    model = planner.kinematics.config.model
    for fe in (planner, planner.trajopt_solver, mpc):
        q = fe.sample_configs(20, rejection_ratio=20)
        assert q.ndim == 2 and q.shape[1] == 7 and 1 <= q.shape[0] <= 20
        qn = q.cpu().numpy()
        lo, hi = model.joint_limits_position
        assert (qn >= lo - 1e-6).all() and (qn <= hi + 1e-6).all()
        fk = oracle.kinematics_forward(qn, model.as_dict())
        sph = fk["robot_spheres"].reshape(len(qn), 1, -1, 4)
        assert (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"].reshape(-1) == 0).all()
    with pytest.raises(ValueError, match="positive"):
        mpc.sample_configs(0)


def test_attached_object_rides_the_tool_frame_and_collides(oracle, device):
    """``AttachmentManager.update`` with the object's WORLD pose (reference attachment_manager.py:105-180): the spheres written into
    the ``attached_object`` slots, carried through FK at the grasp configuration, are the object's spheres in the world; at any
    other configuration they keep their place relative to the tool frame; a world cuboid at the object's place collides with the
    robot only while the object is attached and the cuboid switched on; the planner hands out one manager over its own model"""
    from curobo_amd.attachment_manager import AttachmentManager
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import Kinematics
    from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo_amd.scene.config import scene_from_config
    from curobo_amd.scene.types import Cuboid, Pose7
    from curobo_amd.scene.types import SceneCfg as Scene
    from curobo_amd.types import JointState, Pose

    cfg = _cfg(device)
    kin = Kinematics(cfg, compute_spheres=True)
    kp = kin.kinematics_config
    slots = kp.get_sphere_index_from_link_name("attached_object")
    qg = torch.tensor([[0.0, -1.2, 0.0, -2.0, 0.0, 1.0, 0.0]], device=device)
    hand = kin.get_link_poses(qg, ["panda_hand"])
    hand7 = hand.position[0, 0].tolist() + hand.quaternion[0, 0].tolist()
    # a box held 20 cm along the hand's z axis (clear of the fingers), turned against the hand
    in_hand = Pose7([0.0, 0.0, 0.2, 0.924, 0.383, 0.0, 0.0])
    H = Pose7(hand7)
    box_t = H.transform(in_hand.t[None])[0]
    Rw = H.R @ in_hand.R
    from scipy.spatial.transform import Rotation

    qx, qy, qz, qw = Rotation.from_matrix(Rw).as_quat()
    box_pose = [*box_t.tolist(), float(qw), float(qx), float(qy), float(qz)]
    local = torch.tensor([[0.0, 0.0, 0.0, 0.02], [0.03, 0.0, 0.0, 0.015], [-0.03, 0.01, 0.02, 0.015]], device=device)
    world_pose = Pose(torch.tensor([box_pose[:3]], device=device), torch.tensor([box_pose[3:]], device=device))
    scene = scene_from_config(Scene(cuboid=[Cuboid("held_box", box_pose, dims=[0.08, 0.04, 0.04])]), device)
    m = AttachmentManager(kin, scene)
    m.update(local, JointState.from_position(qg), world_objects_pose_offset=world_pose)
    want = Pose7(box_pose).transform(local[:, :3].cpu().numpy())
    sph = kin.compute_kinematics(qg).robot_spheres[0, 0, slots].cpu().numpy()
    np.testing.assert_allclose(sph[:3, :3], want, atol=2e-6)
    np.testing.assert_allclose(sph[:3, 3], local[:, 3].cpu().numpy(), atol=0)
    assert (sph[3:, 3] < 0).all()
    # another configuration: the same offsets from the hand, by the oracle's FK over the edited model
    q2 = torch.as_tensor(sample_q(cfg.model, 3, seed=5), device=device)
    got = kin.compute_kinematics(q2).robot_spheres[:, 0, slots[:3]].cpu().numpy()
    hands = kin.get_link_poses(q2, ["panda_hand"])
    for b in range(3):
        Hb = Pose7(hands.position[b, 0].tolist() + hands.quaternion[b, 0].tolist())
        np.testing.assert_allclose(got[b, :, :3], Hb.transform(in_hand.transform(local[:, :3].cpu().numpy())), atol=5e-6)
    # the world's copy of the object: in collision with the attached spheres while it is on, clear when the manager switches it off
    chk = RobotCollisionChecker(cfg, scene)
    d_on = chk.get_scene_self_collision_distance_from_joints(qg)[0]
    assert float(d_on.max()) > 0.0
    m.detach()
    assert float(chk.get_scene_self_collision_distance_from_joints(qg)[0].max()) == 0.0  # (the bare hand does not reach the box)
    m.attach_from_scene(JointState.from_position(qg), ["held_box"], num_spheres=4, world_objects_pose_offset=Pose(
        torch.zeros(1, 3, device=device), torch.tensor([[1.0, 0, 0, 0]], device=device)))  # (the fit is in the world frame already)
    n = m._last_fit_result.num_spheres
    sph = kin.compute_kinematics(qg).robot_spheres[0, 0, slots].cpu().numpy()
    np.testing.assert_allclose(sph[:n], torch.cat([m._last_fit_result.centers, m._last_fit_result.radii[:, None]], 1).cpu().numpy(), atol=2e-6)
    assert int(scene.tensors["cuboid_enable"][0, 0]) == 0
    assert float(chk.get_scene_self_collision_distance_from_joints(qg)[0].max()) == 0.0
    m.detach()
    assert int(scene.tensors["cuboid_enable"][0, 0]) == 1
    # front ends: one manager, over the tensors their solvers read
    planner = MotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=Scene(cuboid=[Cuboid("table", [0.7, 0, 0.2, 1, 0, 0, 0], dims=[0.6, 1.0, 0.1])]),
                                                    collision_cache={"cuboid": 4}))
    pm = planner.attachment_manager
    assert pm is planner.trajopt_solver.attachment_manager and pm.kinematics_params is planner.kinematics.config.kinematics_config
    assert pm._scene_collision is planner.trajopt_solver.config.scene and pm._scene_collision.tensors["cuboid_dims"].shape[1] == 4
