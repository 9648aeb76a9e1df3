"""Tool-pose / c-space cost kernels vs the oracle, the IK rollout vs the oracle composition, and
the IK solver end to end."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("project", [0, 1])
def test_tool_pose_distance_kernel(method, project, oracle, device):
    from curobo_amd.backends import cost as Cs

    rng = np.random.default_rng(method * 2 + project)
    b, h, L, G, ngoal = 37, 3, 2, 3, 5

    def rq(*s):
        q = rng.normal(size=s + (4,)).astype(np.float32)
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        return q

    cp, cq = rng.normal(size=(b, h, L, 3)).astype(np.float32), rq(b, h, L)
    gp, gq = rng.normal(size=(ngoal, L, G, 3)).astype(np.float32), rq(ngoal, L, G)
    gq[0, 0, 0] = cq[0, 0, 0]  # an exact match exercises the zero / tolerance branch
    gp[0, 0, 0] = cp[0, 0, 0]
    idx = rng.integers(0, ngoal, size=b).astype(np.int32)
    idx[0] = 0
    w = np.array([100.0, 30.0], np.float32)
    ta, na = rng.uniform(0.5, 1.5, size=(L, 6)).astype(np.float32), rng.uniform(0.0, 1.0, size=(L, 6)).astype(np.float32)
    tt, nt = np.full((L, 2), 1e-4, np.float32), np.full((L, 2), 1e-3, np.float32)
    proj = np.full((L,), project, np.uint8)
    ref = oracle.tool_pose_distance(cp, cq, gp, gq, idx, w, ta, na, tt, nt, proj, method)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    out = dict(distance=torch.zeros(b, h, 2 * L, device=device), position_distance=torch.zeros(b, h, L, device=device),
               rotation_distance=torch.zeros(b, h, L, device=device), position_gradient=torch.zeros(b, h, L, 3, device=device),
               rotation_gradient=torch.zeros(b, h, L, 4, device=device),
               goalset_idx=torch.zeros(b, h, L, dtype=torch.int32, device=device))
    Cs.tool_pose_distance(out["distance"], out["position_distance"], out["rotation_distance"], out["position_gradient"],
                          out["rotation_gradient"], out["goalset_idx"], t(cp), t(cq), t(gp), t(gq), t(idx), t(w), t(ta),
                          t(na), t(tt), t(nt), t(proj), b, h, L, G, method)
    torch.cuda.synchronize()
    assert np.array_equal(out["goalset_idx"].cpu().numpy(), ref["goalset_idx"]), "goal-set indices must be exact"
    for k in ("distance", "position_distance", "rotation_distance", "position_gradient", "rotation_gradient"):
        scale = max(1.0, np.abs(ref[k]).max())
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], atol=1e-5 * scale, rtol=1e-4, err_msg=k)


def test_cspace_position_kernel(oracle, device):
    from curobo_amd.backends import cost as Cs

    model = load_model("franka")
    rng = np.random.default_rng(0)
    b, h, d = 53, 2, 7
    lo, hi = model.joint_limits_position.astype(np.float32)
    pos = (sample_q(model, b * h, seed=1, scale=1.2)).reshape(b, h, d)
    p_b = np.stack([lo, hi])
    cur = sample_q(model, 3, seed=2)
    curv = rng.normal(size=(3, d)).astype(np.float32)
    tgt = sample_q(model, 4, seed=3)
    kw = dict(cspace_target=tgt, cspace_target_idx=rng.integers(0, 4, size=b), cspace_target_weight=2.0,
              cspace_target_dof_weight=rng.uniform(0, 1, size=d).astype(np.float32), squared_l2_reg_weight=(0.5, 0.2),
              current_position=cur, current_velocity=curv, idxs_current_state=rng.integers(0, 3, size=b),
              v_b=np.stack([-model.joint_limits_velocity[1], model.joint_limits_velocity[1]]).astype(np.float32) * 40,
              state_dt=np.array([0.1, 0.0, 0.05], np.float32))
    ref = oracle.cspace_position_cost(pos, p_b, np.array([5000.0, 0.0], np.float32), np.array([0.01, 0.01], np.float32), **kw)
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), device=device) if dt is None else torch.as_tensor(np.ascontiguousarray(a), device=device).to(dt)  # noqa: E731
    oc, og = torch.zeros(b, h, d, device=device), torch.zeros(b, h, d, device=device)
    eff_b = torch.stack([torch.full((d,), -1e9, device=device), torch.full((d,), 1e9, device=device)])
    Cs.cspace_position_cost(oc, og, None, t(pos), None, t(tgt), t(kw["cspace_target_idx"], torch.int32), t(p_b), eff_b,
                            t(np.array([5000.0, 0.0], np.float32)), t(np.array([0.01, 0.01], np.float32)),
                            t(np.array([2.0], np.float32)), t(kw["cspace_target_dof_weight"]),
                            t(np.array([0.5, 0.2], np.float32)), t(cur), t(curv),
                            t(kw["idxs_current_state"], torch.int32), t(kw["v_b"]), t(kw["state_dt"]), True, b, h, d)
    torch.cuda.synchronize()
    assert (ref["cost"] > 0).mean() > 0.3
    np.testing.assert_allclose(oc.cpu().numpy(), ref["cost"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(og.cpu().numpy(), ref["grad_position"], rtol=1e-4, atol=1e-3)


def _ik_setup(device, P=6, S=16):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c1_world

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c1_world())
    scene = SceneData.from_arrays(arrays, device)
    return model, kin, arrays, scene


@pytest.mark.parametrize("fused", [False, True])
def test_ik_rollout_matches_oracle_composition(fused, oracle, device):
    from curobo_amd.rollout.ik_rollout import IKRollout, IKRolloutCfg

    model, kin, arrays, scene = _ik_setup(device)
    md = model.as_dict()
    B = 40
    q = sample_q(model, B, seed=5, scale=1.05)  # a few rows violate the joint limits
    goals = oracle.kinematics_forward(sample_q(model, 4, seed=6, scale=0.7), md)
    ro = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=fused))
    assert ro.fused_available()
    idx = np.arange(B, dtype=np.int32) % 4
    ro.update_goals(torch.as_tensor(goals["link_pos"].reshape(4, 1, 1, 3)), torch.as_tensor(goals["link_quat"].reshape(4, 1, 1, 4)),
                    torch.as_tensor(idx, device=device))
    cost, grad = ro.cost_and_gradient(torch.as_tensor(q, device=device))
    torch.cuda.synchronize()
    c = ro.cfg
    fk = oracle.kinematics_forward(q, md)
    T = 1
    pose = oracle.tool_pose_distance(fk["link_pos"].reshape(B, 1, T, 3), fk["link_quat"].reshape(B, 1, T, 4),
                                     goals["link_pos"].reshape(4, 1, 1, 3), goals["link_quat"].reshape(4, 1, 1, 4), idx,
                                     np.array(c.pose_weight, np.float32), np.ones((T, 6), np.float32), np.ones((T, 6), np.float32),
                                     np.full((T, 2), 1e-8, np.float32), np.full((T, 2), 1e-8, np.float32), np.zeros(T, np.uint8), 0)
    cs = oracle.cspace_position_cost(q.reshape(B, 1, 7), model.joint_limits_position.astype(np.float32),
                                     np.array(c.cspace_weight, np.float32), np.array(c.cspace_activation_distance, np.float32))
    sph = fk["robot_spheres"].reshape(B, 1, -1, 4)
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, c.self_collision_weight)
    wc = oracle.scene_collision(sph, arrays, c.scene_collision_weight, c.scene_activation_distance)
    want = pose["distance"].sum((1, 2)) + cs["cost"].sum((1, 2)) + sc["distance"] + wc["distance"].sum((1, 2))
    assert (sc["distance"] > 0).any() and (wc["distance"] > 0).any() and (cs["cost"] > 0).any()
    np.testing.assert_allclose(cost.cpu().numpy(), want, rtol=1e-4, atol=1e-2)
    gs = sc["gradient"].reshape(B, -1, 4) + wc["gradient"].reshape(B, -1, 4) * np.array([1, 1, 1, 0], np.float32)
    gq = oracle.kinematics_backward(md, fk["cumul_mat"], gs, pose["position_gradient"].reshape(B, 1, 3),
                                    pose["rotation_gradient"].reshape(B, 1, 4)) + cs["grad_position"].reshape(B, 7)
    np.testing.assert_allclose(grad.cpu().numpy(), gq, rtol=2e-3, atol=2e-5 * np.abs(gq).max())


@pytest.mark.parametrize("robot,method,n_extra", [("franka", 0, 0), ("franka", 2, 0), ("unitree_g1", 1, 0), ("franka", 0, 33)])
def test_ik_fused_equals_kernel_sequence(robot, method, n_extra, device):
    """one-launch IK rollout vs the seven drop-in launches: cost, gradient and every metric buffer
    (G1: 4 tool frames, 49 dof, goal sets of 3; the tiled self-collision kernel on the other side)"""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout.ik_rollout import IKRollout, IKRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c1_world

    model = load_model(robot)
    kin = KinematicsParams.from_model(model, device)
    rng = np.random.default_rng(9)
    world = c1_world()
    for _ in range(n_extra):  # > 32 obstacle records: beyond the 32-bit link masks (never culled)
        p = rng.uniform([-0.6, -0.6, 0.1], [0.7, 0.7, 0.9])
        world[0].append({"dims": rng.uniform(0.04, 0.12, 3).tolist(), "pose": [*p.tolist(), 1, 0, 0, 0]})
    scene = SceneData.from_arrays(cuboid_scene_arrays(world), device)
    B, T, G, NG = 53, kin.num_pose_links, 3, 5
    q = torch.as_tensor(sample_q(model, B, seed=5, scale=1.05), device=device)
    gpos = torch.as_tensor(rng.normal(size=(NG, T, G, 3)).astype(np.float32) * 0.5, device=device)
    gq = rng.normal(size=(NG, T, G, 4)).astype(np.float32)
    gq /= np.linalg.norm(gq, axis=-1, keepdims=True)
    idx = torch.as_tensor(rng.integers(0, NG, size=B).astype(np.int32), device=device)
    outs = []
    for fused in (False, True):
        ro = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=fused, rotation_method=method), num_goalset=G)
        ro.update_goals(gpos, torch.as_tensor(gq, device=device), idx)
        if fused:
            if not ro.fused_available():
                pytest.skip("16 configurations of this robot do not fit in LDS")
            cost, grad = ro.cost_and_gradient_fused(q, with_metrics=True)
        else:
            cost, grad = ro.cost_and_gradient(q)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (cost, grad, ro.pose_cost, ro.pose_pos_dist, ro.pose_rot_dist, ro.goalset_idx,
                                         ro.link_pos, ro.link_quat, ro.robot_spheres, ro.cspace_cost)])
    ref, got = outs
    assert torch.equal(ref[5], got[5]), "goal-set indices must be exact"
    assert float(ref[0].max()) > 0
    for i, (a_, b_) in enumerate(zip(ref, got)):
        if i == 5:
            continue
        tol = dict(rtol=2e-3, atol=2e-5 * float(a_.abs().max())) if i == 1 else dict(rtol=2e-5, atol=2e-5 * max(1.0, float(a_.abs().max())))
        torch.testing.assert_close(b_.reshape(a_.shape), a_, **tol)


@pytest.mark.parametrize("exit_early", [False, True])
def test_ik_solver_end_to_end(exit_early, oracle, device):
    from curobo_amd.solver import IKSolver, IKSolverCfg

    model, kin, arrays, scene = _ik_setup(device)
    md = model.as_dict()
    P = 12
    # reachable, collision-free goals: FK of joint samples that the oracle finds collision free
    cand = sample_q(model, 400, seed=11, scale=0.8)
    fk = oracle.kinematics_forward(cand, md)
    sph = fk["robot_spheres"].reshape(400, 1, -1, 4)
    free = (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0) & \
        (oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0)
    sel = np.nonzero(free)[0][:P]
    assert len(sel) == P
    gp, gq = fk["link_pos"][sel, 0], fk["link_quat"][sel, 0]
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=32, exit_early=exit_early))
    res = solver.solve_pose(torch.as_tensor(gp), torch.as_tensor(gq))
    torch.cuda.synchronize()
    succ = res.success.cpu().numpy()
    assert succ.mean() >= 0.9, f"IK success rate {succ.mean():.2f}"
    # reference exit_early (solver_ik.py:395-404): the L-BFGS stage is skipped exactly when the seed-IK
    # solutions already solve every problem; the checks below hold for either kind of solution
    if not exit_early:
        assert solver.optimizer_ran
    elif not solver.optimizer_ran:
        assert succ.all()
    # verify the reported solutions with the oracle: pose reached, limits respected, collision free
    qs = res.solution.cpu().numpy()[succ]
    chk = oracle.kinematics_forward(qs, md)
    np.testing.assert_allclose(chk["link_pos"][:, 0], gp[succ], atol=5e-3)
    dotq = np.abs((chk["link_quat"][:, 0] * gq[succ]).sum(-1))
    assert (2 * np.arccos(np.clip(dotq, 0, 1)) < 0.05).all()
    lo, hi = model.joint_limits_position
    assert (qs >= lo - 1e-4).all() and (qs <= hi + 1e-4).all()
    s2 = chk["robot_spheres"].reshape(len(qs), 1, -1, 4)
    assert (oracle.self_collision(s2, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all()
    assert (oracle.scene_collision(s2, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0).all()


def test_ik_solver_goalset(oracle, device):
    """Goal sets (reference solve_pose with num_goalset > 1): member 0 of every problem is out of
    reach, so a solution has to pick one of the two reachable members and say which."""
    from curobo_amd.solver import IKSolver, IKSolverCfg

    model, kin, arrays, scene = _ik_setup(device)
    md = model.as_dict()
    P, G = 8, 3
    cand = sample_q(model, 600, seed=21, scale=0.8)
    fk = oracle.kinematics_forward(cand, md)
    sph = fk["robot_spheres"].reshape(600, 1, -1, 4)
    free = (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0) & \
        (oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0)
    sel = np.nonzero(free)[0][:P * (G - 1)]
    assert len(sel) == P * (G - 1)
    gp = np.zeros((P, G, 3), np.float32)
    gq = np.zeros((P, G, 4), np.float32)
    gp[:, 0] = [3.0, 0.0, 0.5]  # 3 m away: unreachable
    gq[:, 0] = [1, 0, 0, 0]
    gp[:, 1:] = fk["link_pos"][sel, 0].reshape(P, G - 1, 3)
    gq[:, 1:] = fk["link_quat"][sel, 0].reshape(P, G - 1, 4)
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=32, num_goalset=G))
    res = solver.solve_pose(torch.as_tensor(gp), torch.as_tensor(gq))
    torch.cuda.synchronize()
    succ = res.success.cpu().numpy()
    assert succ.mean() >= 0.85, f"goal-set IK success rate {succ.mean():.2f}"
    assert res.goalset_index.shape == (P, 1), "one member index per tool frame"
    member = res.goalset_index[:, 0].cpu().numpy()
    assert set(np.unique(member[succ])) <= {1, 2}
    qs = res.solution.cpu().numpy()[succ]
    chk = oracle.kinematics_forward(qs, md)
    reached_p = gp[np.arange(P), member][succ]
    reached_q = gq[np.arange(P), member][succ]
    np.testing.assert_allclose(chk["link_pos"][:, 0], reached_p, atol=5e-3)
    dotq = np.abs((chk["link_quat"][:, 0] * reached_q).sum(-1))
    assert (2 * np.arccos(np.clip(dotq, 0, 1)) < 0.05).all()
    # the top-k interface returns distinct ranked seeds per problem, best first
    top = solver.solve_pose(torch.as_tensor(gp), torch.as_tensor(gq), return_seeds=4)
    assert top.solution.shape == (P, 4, kin.num_dof) and top.goalset_index.shape == (P, 4, 1)
    c = top.cost.cpu().numpy() + 1e16 * (~top.success.cpu().numpy())
    assert (np.diff(c, axis=1) >= 0).all()
    assert top.success[:, 0].float().mean().item() >= 0.85  # (the seed sampler advances between solves: not the same seeds)
    with pytest.raises(ValueError, match="goal-set members"):
        solver.solve_pose(torch.as_tensor(gp[:, 0]), torch.as_tensor(gq[:, 0]))


@pytest.mark.parametrize("retime", [False, True])
def test_cspace_state_kernel(retime, oracle, device):
    from curobo_amd.backends import cost as Cs

    rng = np.random.default_rng(3)
    b, h, d = 41, 9, 7
    x = {k: rng.normal(size=(b, h, d)).astype(np.float32) * s for k, s in
         (("pos", 2.0), ("vel", 3.0), ("acc", 8.0), ("jerk", 30.0), ("effort", 40.0))}
    lim = {k: np.stack([-np.ones(d), np.ones(d)]).astype(np.float32) * s for k, s in
           (("position", 1.5), ("velocity", 2.0), ("acceleration", 6.0), ("jerk", 25.0), ("effort", 30.0))}
    dt = rng.uniform(0.02, 0.1, size=b).astype(np.float32)
    tgt = rng.normal(size=(3, d)).astype(np.float32)
    tidx = rng.integers(0, 3, size=b).astype(np.int32)
    dofw = rng.uniform(0.5, 1, size=d).astype(np.float32)
    w, eta, reg = [50.0, 20.0, 5.0, 1.0, 2.0], [0.05, 0.1, 0.1, 0.1, 0.1], [0.3, 0.2, 0.1, 0.05, 0.4]
    ref = oracle.cspace_state_cost(x["pos"], x["vel"], x["acc"], x["jerk"], dt, lim, w, eta, reg, effort=x["effort"], target=tgt,
                                   idxs_target=tidx, target_weight=3.0, non_terminal_factor=0.25, target_dof_weight=dofw,
                                   retime_weights=retime, retime_regularization_weights=retime)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32 if np.asarray(a).dtype.kind == "f" else None), device=device)  # noqa: E731
    outs = [torch.zeros(b, h, d, device=device) for _ in range(6)]
    Cs.cspace_state_cost(*outs, t(x["pos"]), t(x["vel"]), t(x["acc"]), t(x["jerk"]), t(x["effort"]), t(dt), t(tgt), t(tidx),
                         t(lim["position"]), t(lim["velocity"]), t(lim["acceleration"]), t(lim["jerk"]), t(lim["effort"]),
                         t(np.array(w, np.float32)), t(np.array(eta, np.float32)), t(np.array(reg, np.float32)),
                         t(np.array([3.0], np.float32)), t(np.array([0.25], np.float32)), t(dofw), True, b, h, d, retime, retime)
    torch.cuda.synchronize()
    for o, k in zip(outs, ("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")):
        np.testing.assert_allclose(o.cpu().numpy(), ref[k], rtol=2e-5, atol=2e-5 * max(1.0, np.abs(ref[k]).max()), err_msg=k)


def test_ik_solver_stream_shards_give_the_same_solutions(device):
    """Problem shards on separate HIP streams (IKSolverCfg.stream_shards) are the same optimisation:
    identical winners and errors as the one-stream solver from the same seeds."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import IKSolver, IKSolverCfg
    from curobo_amd.workloads import c1_world, reachable_goals

    kin = KinematicsParams.from_model(load_model("franka"), device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), device)
    P, S = 8, 16
    gp, gq = reachable_goals(kin, P, seed=3)
    ref = None
    for shards in (1, 2, 4):
        solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, stream_shards=shards, use_lm_seed=False))
        res = solver.solve_pose(gp, gq, seeds=solver.sample_seeds())
        torch.cuda.synchronize()
        if ref is None:
            ref = res
            assert res.success.float().mean() > 0.5
        else:
            assert torch.equal(res.success, ref.success) and torch.equal(res.seed_index, ref.seed_index)
            assert torch.equal(res.solution, ref.solution)


@pytest.mark.parametrize("aligned", [True, False])
def test_ik_fused_multi_env_equals_kernel_sequence(aligned, device):
    """batch-env IK rollout: every problem has its own world (cuboids + an ESDF grid) and its own collision-sphere set; the
    one-launch form serves a workgroup's 16 configurations from one staged environment, so it runs when ``env_query_idx``
    is constant over aligned runs of 16 rows (seeds of one problem) and hands over to the kernel sequence when it is not"""
    import dataclasses

    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout.ik_rollout import IKRollout, IKRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c5_mixed_worlds

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    n_env = 3
    sph = kin.link_spheres.repeat(n_env, 1, 1).clone()  # per-environment sphere sets: radii differ
    sph[1, :, 3] = torch.where(sph[1, :, 3] > 0, sph[1, :, 3] * 1.3, sph[1, :, 3])
    sph[2, :, :3] += 0.01
    kin = dataclasses.replace(kin, link_spheres=sph.contiguous())
    assert kin.num_envs == n_env
    scene = SceneData.from_arrays(c5_mixed_worlds(n_env, grid=32), device)
    seeds = 32 if aligned else 20  # 20 seeds per problem: runs of 16 rows straddle two problems
    B = n_env * seeds
    env = torch.arange(n_env, dtype=torch.int32, device=device).repeat_interleave(seeds)
    rng = np.random.default_rng(3)
    q = torch.as_tensor(sample_q(model, B, seed=8, scale=0.9), device=device)
    T = kin.num_pose_links
    gpos = torch.as_tensor(rng.normal(size=(n_env, T, 1, 3)).astype(np.float32) * 0.4, device=device)
    gq = rng.normal(size=(n_env, T, 1, 4)).astype(np.float32)
    gq /= np.linalg.norm(gq, axis=-1, keepdims=True)
    idx = env.clone()
    outs = []
    for fused in (False, True):
        ro = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=fused), num_goalset=1)
        ro.update_goals(gpos, torch.as_tensor(gq, device=device), idx)
        ro.update_env_query_idx(env)
        assert ro._env_runs_ok == aligned
        cost, grad = ro.cost_and_gradient(q)
        torch.cuda.synchronize()
        outs.append((cost.clone(), grad.clone(), ro))
    (c0, g0, _), (c1, g1, ro1) = outs
    assert ro1.fused_available()
    assert float(c0.max()) > 0 and float((c0 > 0).float().mean()) > 0.3
    torch.testing.assert_close(c1, c0, rtol=2e-5, atol=1e-3)
    torch.testing.assert_close(g1, g0, rtol=1e-3, atol=2e-5 * float(g0.abs().max()))
    # the environments matter: the same rows against environment 0 only cost something else
    ro0 = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=False), num_goalset=1)
    ro0.update_goals(gpos, torch.as_tensor(gq, device=device), idx)
    ro0.update_env_query_idx(torch.zeros_like(env))
    cz, _ = ro0.cost_and_gradient(q)
    torch.cuda.synchronize()
    assert float((cz - c0).abs().max()) > 1e-3


def test_front_end_takes_the_robots_configuration(device):
    """``InverseKinematics.solve_pose(current_state=)`` (reference solver_ik.py:631-700): the configuration is the first seed of the
    seed stage and its ranking prefers solutions near it -- a goal the robot is already AT comes back as the configuration it is
    in, where without it the solver returns whichever of the arm's solutions ranks first"""
    from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
    from curobo_amd.types import JointState

    n = 20
    ik = InverseKinematics(InverseKinematicsCfg.create(robot="franka.yml", scene_model="collision_table.yml", num_seeds=8, max_batch_size=n))
    q = ik.sample_configs(n, rejection_ratio=30)[:n].contiguous()
    assert q.shape[0] == n
    goal = ik.compute_kinematics(JointState.from_position(q)).tool_poses.as_goal()
    ik.reset_seed()
    near = ik.solve_pose(goal, current_state=JointState.from_position(q))
    ik.reset_seed()
    free = ik.solve_pose(goal)
    assert bool(near.success.all()) and bool(free.success.all())
    d_near = (near.solution[:, 0] - q).abs().max(-1).values
    d_free = (free.solution[:, 0] - q).abs().max(-1).values
    assert float(d_near.max()) < 2e-2, d_near
    assert float(d_free.max()) > 0.1  # (some other branch of the arm's solutions is returned for at least one problem)
