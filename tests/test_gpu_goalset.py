"""Goal sets and per-axis pose criteria through the trajectory-optimisation rollout and the planners, and the grasp plan
built from them: reference ``cost/wp_tool_pose.py:456-692`` (closest member of a goal set, terminal / non-terminal axis
factors, distance projected into the goal frame), ``cost/tool_pose_criteria.py:17-200``, ``motion/motion_planner.py:419-640``
(``plan_grasp``, ``enable_link_collision`` / ``disable_link_collision``, ``update_tool_pose_criteria``,
``update_link_inertial``).  Pose terms are checked against the oracle's tool-pose function on the rollout's own link poses,
planned motions against the oracle's forward kinematics."""

import dataclasses

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q
from test_gpu_planner import _scene_cfg, _verify_with_oracle, this_repos_curobo  # noqa: F401

pytestmark = pytest.mark.gpu


def _quat_rotate_np(q, v):
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


@pytest.mark.parametrize("project", [False, True])
def test_trajopt_rollout_goal_set_and_criteria_match_oracle(project, oracle, device):
    """three goal poses per goal row, linear-motion criteria (the points before the last one are scored on two position axes
    and all rotations, optionally in the goal frame): kernel sequence and fused launch report the oracle's pose cost,
    distance and chosen member on their own link poses, and agree on cost and gradient"""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.types import ToolPoseCriteria
    from curobo_amd.workloads import c1_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), device)
    md = model.as_dict()
    B, G, NG = 12, 3, 3
    knots = torch.as_tensor(seed_knots(model, B, 12, seed=5, spread=0.6), device=device)
    start = torch.as_tensor(start_configuration(model), device=device)
    fk = oracle.kinematics_forward(sample_q(model, G * NG, seed=9, scale=0.6), md)
    gpos = fk["link_pos"].reshape(G, 1, NG, 3).astype(np.float32)
    gquat = fk["link_quat"].reshape(G, 1, NG, 4).astype(np.float32)
    idx = (np.arange(B) % G).astype(np.int32)
    crit = ToolPoseCriteria.linear_motion("z", non_terminal_scale=0.5, project_distance_to_goal=project)
    frame = kin.tool_frames[0]
    out = []
    for fused in (False, True):
        ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(use_fused=fused, use_sweep=False, use_speed_metric=False, traj_dt=0.1))
        ro.update_start_state(start)
        ro.update_goals(torch.as_tensor(gpos, device=device), torch.as_tensor(gquat, device=device), torch.as_tensor(idx, device=device))
        ro.update_tool_pose_criteria({frame: crit})
        if fused:
            assert ro.fused_available()
            c, g = [t.clone() for t in ro.cost_and_gradient_fused(knots, with_metrics=True)]
        else:
            c = ro.evaluate_action(knots, with_gradient=True).clone()
            g = ro.grad_knots.view(B, -1).clone()
        torch.cuda.synchronize()
        H = ro.position.shape[1]
        if fused:  # the fused launch materialises the joint trajectory, not the link poses: the oracle's FK of it
            lk = oracle.kinematics_forward(ro.position.cpu().numpy().reshape(B * H, -1), md, horizon=H)
            link_pos, link_quat = lk["link_pos"].reshape(B, H, 1, 3), lk["link_quat"].reshape(B, H, 1, 4)
        else:
            link_pos, link_quat = ro.link_pos.cpu().numpy(), ro.link_quat.cpu().numpy()
        want = oracle.tool_pose_distance(
            link_pos, link_quat, gpos, gquat, idx, np.array(ro.cfg.pose_weight, np.float32),
            np.array([crit.terminal_pose_axes_weight_factor], np.float32), np.array([crit.non_terminal_pose_axes_weight_factor], np.float32),
            np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32), np.array([int(project)], np.uint8), 0)
        got = ro.pose_cost.cpu().numpy().reshape(B, H, 2)
        assert (want["distance"][:, :-1] > 0).any(), "the non-terminal factors act"
        np.testing.assert_allclose(got, want["distance"], rtol=2e-4, atol=2e-5 * want["distance"].max())
        np.testing.assert_allclose(ro.pose_pos_dist.cpu().numpy().reshape(B, H, 1), want["position_distance"], rtol=2e-4, atol=1e-5)
        member = ro.goalset_idx.cpu().numpy().reshape(B, H, 1)
        # (members that tie to within rounding may swap; none do on this data)
        assert (member == want["goalset_idx"]).mean() > 0.99
        assert len(np.unique(member)) == NG, "every member of the set is the closest one somewhere"
        out.append((c, g))
    (c0, g0), (c1, g1) = out
    torch.testing.assert_close(c1, c0, rtol=2e-4, atol=1e-1)
    torch.testing.assert_close(g1, g0, rtol=2e-3, atol=5e-5 * float(g0.abs().max()))
    # the standard criteria put the rollout back: pose cost on the last point only
    ro.update_tool_pose_criteria({frame: ToolPoseCriteria()})
    ro.cost_and_gradient_fused(knots, with_metrics=True)
    torch.cuda.synchronize()
    assert float(ro.pose_cost.view(B, H, 2)[:, :-1].abs().max()) == 0.0 and float(ro.pose_cost.view(B, H, 2)[:, -1].sum()) > 0


def test_link_collision_toggle_and_inertial_updates(oracle, device):
    """``disable_link_spheres`` takes a link out of the self and scene collision terms (radius -100 in the sphere table the
    kernels read), ``enable_link_spheres`` puts it back bit for bit; ``update_link_inertial`` rewrites the rows the RNEA reads"""
    from curobo_amd.dynamics import Dynamics
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout.ik_rollout import IKRollout, IKRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    md = model.as_dict()
    B, D = 32, model.num_dof
    q0 = np.asarray(model.cspace["default_joint_position"], np.float32)[:D]
    q = (q0[None] + 0.1 * sample_q(model, B, seed=3, scale=0.3)).astype(np.float32)
    hand, _ = _tool_pose(oracle, model, q0)
    world = [[{"dims": [0.25, 0.25, 0.25], "pose": [float(hand[0, 0]), float(hand[0, 1]), float(hand[0, 2]), 1, 0, 0, 0]}]]
    arrays = cuboid_scene_arrays(world)
    scene = SceneData.from_arrays(arrays, device)
    seq = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=False))
    fz = IKRollout(kin, scene, B, IKRolloutCfg(use_fused=True))
    goal = torch.as_tensor(hand.reshape(1, 1, 1, 3), device=device)
    gq = torch.tensor([1.0, 0, 0, 0], device=device).view(1, 1, 1, 4)
    for ro in (seq, fz):
        ro.update_goals(goal, gq, torch.zeros(B, dtype=torch.int32, device=device))
    qd = torch.as_tensor(q, device=device)

    def terms():
        seq.evaluate(qd, with_gradient=False)
        assert fz.fused_available()
        c = fz.cost_and_gradient_fused(qd)[0].clone()
        torch.cuda.synchronize()
        return seq.self_dist.view(B).cpu().numpy().copy(), seq.scene_dist.view(B, -1).cpu().numpy().copy(), c.cpu().numpy()

    def oracle_terms(m):
        sph = oracle.kinematics_forward(q, m)["robot_spheres"].reshape(B, 1, -1, 4)
        sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, seq.cfg.self_collision_weight)
        wc = oracle.scene_collision(sph, arrays, seq.cfg.scene_collision_weight, seq.cfg.scene_activation_distance)
        return sc["distance"].reshape(B), wc["distance"].reshape(B, -1)

    s_on, w_on, c_on = terms()
    so, wo = oracle_terms(md)
    np.testing.assert_allclose(s_on, so, rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(w_on, wo, rtol=2e-4, atol=1e-3)
    assert (w_on.sum(1) > 0).all(), "the hand is inside the box"
    links = ["panda_hand", "panda_leftfinger", "panda_rightfinger"]
    idx = torch.cat([kin.get_sphere_index_from_link_name(n) for n in links]).cpu().numpy()
    for n in links:
        kin.disable_link_spheres(n)
    s_off, w_off, c_off = terms()
    md_off = dict(md)
    md_off["link_spheres"] = kin.link_spheres.cpu().numpy()
    so, wo = oracle_terms(md_off)
    np.testing.assert_allclose(s_off, so, rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(w_off, wo, rtol=2e-4, atol=1e-3)
    assert (w_off[:, idx] == 0).all() and (w_off.sum(1) < w_on.sum(1)).all()
    # the fused launch reads the same table: its cost drops by what the sequence's collision terms drop by
    np.testing.assert_allclose(c_on - c_off, (s_on + w_on.sum(1)) - (s_off + w_off.sum(1)), rtol=1e-3, atol=1e-2)
    for n in links:
        kin.enable_link_spheres(n)
    assert torch.equal(kin.link_spheres, kin.reference_link_spheres)
    s2, w2, c2 = terms()
    np.testing.assert_array_equal(w2, w_on)
    np.testing.assert_array_equal(c2, c_on)
    with pytest.raises(ValueError, match="not found"):
        kin.disable_link_spheres("no_such_link")

    # inertial update: the gravity torques change as the oracle's RNEA says
    n = 8
    qq = sample_q(model, n, seed=1, scale=0.5)
    z = np.zeros_like(qq)
    dyn = Dynamics(kin)
    tz = torch.zeros(n, D, device=device)

    def torques():
        tau = dyn.compute_inverse_dynamics(torch.as_tensor(qq, device=device), tz, tz)
        torch.cuda.synchronize()
        return tau.detach().cpu().numpy().reshape(n, -1)

    t0 = torques()
    kin.update_link_inertial("panda_hand", mass=3.0, com=[0.0, 0.02, 0.08], inertia=[0.02, 0.02, 0.01, 0.0, 0.0, 0.0])
    md2 = dict(md)
    md2["link_masses_com"], md2["link_inertias"] = kin.link_masses_com.cpu().numpy(), kin.link_inertias.cpu().numpy()
    k = model.link_names.index("panda_hand")
    np.testing.assert_allclose(md2["link_masses_com"][k], [0.0, 0.02, 0.08, 3.0], atol=1e-7)
    np.testing.assert_allclose(md2["link_inertias"][k, :6], [0.02, 0.02, 0.01, 0.0, 0.0, 0.0], atol=1e-7)
    want, _ = oracle.rnea_forward(qq, z, z, md2)
    t1 = torques()
    np.testing.assert_allclose(t1, want.reshape(n, -1), rtol=2e-4, atol=2e-4)
    assert np.abs(t1 - t0).max() > 1.0
    kin.update_links_inertial({"panda_hand": {"mass": float(model.link_masses_com[k, 3]), "com": model.link_masses_com[k, :3],
                                              "inertia": model.link_inertias[k, :6]}})
    np.testing.assert_allclose(torques(), t0, rtol=0, atol=0)
    with pytest.raises(ValueError, match="At least one"):
        kin.update_link_inertial("panda_hand")


def _tool_pose(oracle, model, q):
    fk = oracle.kinematics_forward(np.ascontiguousarray(q, np.float32).reshape(-1, model.num_dof), model.as_dict())
    return fk["link_pos"].reshape(-1, 3), fk["link_quat"].reshape(-1, 4)


def test_planner_goal_set_and_plan_grasp(oracle, device, this_repos_curobo):  # noqa: F811
    """``plan_pose`` on a goal set reaches one member and says which; ``plan_grasp`` = goal-set plan -> approach pose ->
    straight line to the grasp -> straight-line lift.  Every leg is verified with the oracle: feasible, ends in its pose,
    and the two linear legs stay on their line at the grasp orientation."""
    from curobo.motion_planner import GraspPlanResult, MotionPlanner, MotionPlannerCfg
    from curobo.types import GoalToolPose, JointState, Pose
    from curobo_amd.scene import cuboid_scene_arrays

    world = [{"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]},
             {"dims": [0.2, 0.2, 0.5], "pose": [0.2, 0.6, 0.25, 1, 0, 0, 0]}]
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=_scene_cfg(world), num_ik_seeds=32, num_trajopt_seeds=4,
                                     max_goalset=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    kp = config.trajopt_solver_config.kinematics.kinematics_config
    arrays = cuboid_scene_arrays([world])
    rc = config.trajopt_solver_config.solver_cfg().rollout
    cur = JointState.from_position(planner.default_joint_state.position.view(1, -1).clone(), planner.joint_names)
    q0 = cur.position[0].cpu().numpy()
    frame = planner.tool_frames[0]
    # grasp candidates: three poses the arm reaches (FK of configurations near the default one) and one 3 m away
    dq = np.array([[0.5, 0.25, 0, -0.2, 0, 0.3, 0], [-0.4, 0.3, 0, -0.1, 0, 0.2, 0.3], [0.1, 0.4, 0.2, 0.1, 0, 0.4, -0.3]], np.float32)
    cp, cq = _tool_pose(oracle, model, q0[None] + dq)
    cand_p = np.concatenate([np.array([[3.0, 0.0, 0.5]], np.float32), cp]).astype(np.float32)
    cand_q = np.concatenate([cq[:1], cq]).astype(np.float32)
    grasps = GoalToolPose.from_poses({frame: Pose(torch.as_tensor(cand_p, device=device), torch.as_tensor(cand_q, device=device))},
                                     num_goalset=4)
    assert grasps.position.shape == (1, 1, 1, 4, 3) and grasps.num_goalset == 4

    res = planner.plan_pose(grasps, cur)
    assert res is not None and bool(res.success[0, 0])
    gi = int(res.goalset_index[0, 0])
    assert gi in (1, 2, 3)
    traj = res.js_solution.position[0].cpu().numpy()
    _verify_with_oracle(oracle, model, arrays, traj, res.js_solution.dt[0].cpu().numpy(), q0, rc)
    endp, _ = _tool_pose(oracle, model, traj[0, -1])
    np.testing.assert_allclose(endp[0], cand_p[gi], atol=5e-3)
    # a smaller set than the planner was built for is padded: same call surface
    one = planner.plan_pose(GoalToolPose.from_poses({frame: Pose(torch.as_tensor(cand_p[2:3], device=device),
                                                                 torch.as_tensor(cand_q[2:3], device=device))}), cur)
    assert bool(one.success[0, 0]) and int(one.goalset_index[0, 0]) == 0
    with pytest.raises(ValueError, match="max_goalset"):
        big = GoalToolPose(grasps.tool_frames, grasps.position.repeat(1, 1, 1, 2, 1), grasps.quaternion.repeat(1, 1, 1, 2, 1))
        planner.trajopt_solver.solve_pose(big, cur, seed_config=cur.position.view(1, 1, -1).repeat(1, 4, 1))

    # ---- grasp plan
    offset = -0.10
    g = planner.plan_grasp(grasps, cur, grasp_approach_offset=offset, grasp_lift_offset=offset)
    assert isinstance(g, GraspPlanResult)
    assert bool(g.success.all()) and bool(g.approach_success.all()) and bool(g.grasp_success.all()) and bool(g.lift_success.all()), g.status
    assert g.status == "Planning to lift pose succeeded."
    gi = int(g.goalset_index.view(-1)[0])
    assert gi in (1, 2, 3)
    # the toggles and the criteria are back where they were
    assert torch.equal(kp.link_spheres, kp.reference_link_spheres)
    ro = planner.trajopt_solver.solver.rollout
    assert float(ro._axes_w0.abs().max()) == 0.0 and int(ro._project.max()) == 0
    gp, gq = cand_p[gi], cand_q[gi]
    axis = _quat_rotate_np(gq, np.array([0.0, 0.0, 1.0], np.float32))  # tool z in the world
    approach_p = gp + offset * axis
    legs = [("approach", g.approach_trajectory, q0, approach_p, False), ("grasp", g.grasp_trajectory, None, gp, True),
            ("lift", g.lift_trajectory, None, approach_p, True)]
    prev_end = None
    for name, js, start, goal_p, linear in legs:
        traj = js.position[0].cpu().numpy()  # [1, H, D]
        start = prev_end if start is None else start
        if linear:  # contact links' spheres are off on these legs: check the rest of the robot
            for n in kp.grasp_contact_link_names or []:
                if n in kp.link_names:
                    kp.disable_link_spheres(n)
            m2 = dataclasses.replace(model, link_spheres=kp.link_spheres.cpu().numpy())
            for n in kp.grasp_contact_link_names or []:
                if n in kp.link_names:
                    kp.enable_link_spheres(n)
        else:
            m2 = model
        _verify_with_oracle(oracle, m2, arrays, traj, js.dt[0].cpu().numpy(), start, rc)
        p, qn = _tool_pose(oracle, model, traj[0])
        np.testing.assert_allclose(p[-1], goal_p, atol=5e-3, err_msg=name)
        assert min(np.abs(qn[-1] - gq).max(), np.abs(qn[-1] + gq).max()) < 2e-2, name
        if linear:
            d = p - gp
            lateral = d - (d @ axis)[:, None] * axis
            assert np.linalg.norm(lateral, axis=1).max() < 0.01, (name, np.linalg.norm(lateral, axis=1).max())
            ang = np.minimum(np.linalg.norm(qn - gq, axis=1), np.linalg.norm(qn + gq, axis=1))
            assert ang.max() < 0.05, (name, ang.max())
            along = d @ axis
            assert along.min() > offset - 0.01 and along.max() < 0.01
        prev_end = traj[0, -1]
    assert torch.equal(kp.link_spheres, kp.reference_link_spheres)
    # approach only
    g2 = planner.plan_grasp(grasps, cur, grasp_approach_offset=offset, plan_approach_to_grasp=False)
    assert bool(g2.success.all()) and g2.grasp_trajectory is None and g2.status == "Planning to approach pose succeeded."
    # nothing reachable
    far = GoalToolPose(grasps.tool_frames, grasps.position + torch.tensor([3.0, 0, 0], device=device), grasps.quaternion.clone())
    g3 = planner.plan_grasp(far, cur)
    assert not bool(g3.success.any()) and g3.status in ("Goalset planning returned None.", "No grasp in goal set was reachable.")


def test_ik_position_only_criteria(oracle, device, this_repos_curobo):  # noqa: F811
    """``InverseKinematics.update_tool_pose_criteria`` with ``ToolPoseCriteria.track_position``: the orientation of the goal is
    ignored (it is a rotation the arm cannot take there: the goal orientation flipped), the position is reached; the
    standard criteria then make the same goals fail on rotation"""
    from curobo.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
    from curobo.types import GoalToolPose, JointState, ToolPoseCriteria

    n = 16
    config = InverseKinematicsCfg.create(robot="franka.yml", scene_model="collision_table.yml", num_seeds=16, max_batch_size=n)
    ik = InverseKinematics(config)
    model = config.kinematics.model
    frame = ik.tool_frames[0]
    q = ik.sample_configs(n)
    goal = ik.compute_kinematics(JointState.from_position(q)).tool_poses.as_goal()
    # same positions, orientations of OTHER samples: mostly unreachable as a full pose
    mixed = GoalToolPose(goal.tool_frames, goal.position.clone(), goal.quaternion.flip(0).contiguous())
    full = ik.solve_pose(mixed)
    ik.update_tool_pose_criteria({frame: ToolPoseCriteria.track_position()})
    ik.reset_seed()
    pos_only = ik.solve_pose(mixed)
    assert int(pos_only.success.sum()) == n and int(pos_only.success.sum()) >= int(full.success.sum())
    sol = pos_only.solution[:, 0].cpu().numpy()
    p, qn = _tool_pose(oracle, model, sol)
    np.testing.assert_allclose(p, mixed.position[:, 0, 0, 0].cpu().numpy(), atol=5e-3)
    gq = mixed.quaternion[:, 0, 0, 0].cpu().numpy()
    off = np.minimum(np.linalg.norm(qn - gq, axis=1), np.linalg.norm(qn + gq, axis=1))
    # where the full pose cannot be reached the position-only solution leaves the goal's orientation (where it can, the seed stage's
    # full-pose solution is a valid position-only solution too and may be returned as it is)
    unreachable = ~full.success[:, 0].cpu().numpy()
    assert unreachable.sum() >= 3 and (off[unreachable] > 0.02).all(), ("the orientation was free", unreachable, off)
    fk = oracle.kinematics_forward(sol, model.as_dict())
    sph = fk["robot_spheres"].reshape(n, 1, -1, 4)
    assert (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all()
    # back to the standard criteria: a solver built before and one built after the call both see them
    ik.update_tool_pose_criteria({frame: ToolPoseCriteria()})
    ik.reset_seed()
    again = ik.solve_pose(mixed)
    ok = again.success[:, 0].cpu().numpy()
    assert (again.rotation_error[:, 0].cpu().numpy()[ok] < 0.05).all()
    np.testing.assert_array_equal(ok, full.success[:, 0].cpu().numpy())


def test_batch_planner_plan_grasp(oracle, device, this_repos_curobo):  # noqa: F811
    """``BatchMotionPlanner.plan_grasp`` (reference motion_planner_batch.py:291-472): three problems with their own grasp sets,
    one of them out of reach -- its flags stay off and the others' legs are verified with the oracle like the single planner's"""
    from curobo.batch_motion_planner import BatchMotionPlanner, MotionPlannerCfg
    from curobo.types import GoalToolPose, JointState, Pose
    from curobo_amd.scene import cuboid_scene_arrays

    world = [{"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}]
    n, G = 3, 2
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=_scene_cfg(world), num_ik_seeds=32, num_trajopt_seeds=4,
                                     max_batch_size=n, max_goalset=G)
    planner = BatchMotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    kp = config.trajopt_solver_config.kinematics.kinematics_config
    arrays = cuboid_scene_arrays([world])
    rc = config.trajopt_solver_config.solver_cfg().rollout
    q0 = planner.default_joint_state.position.view(1, -1).repeat(n, 1)
    cur = JointState.from_position(q0.clone(), planner.joint_names)
    q0n = q0.cpu().numpy()
    frame = planner.tool_frames[0]
    dq = np.array([[0.5, 0.25, 0, -0.2, 0, 0.3, 0], [-0.4, 0.3, 0, -0.1, 0, 0.2, 0.3], [0.1, 0.4, 0.2, 0.1, 0, 0.4, -0.3],
                   [0.3, 0.1, -0.2, 0.2, 0, 0.1, 0.2]], np.float32)
    p, q = _tool_pose(oracle, model, q0n[:1] + dq)
    cand_p = np.stack([p[[0, 1]], p[[2, 3]], p[[0, 1]] + np.array([3.0, 0, 0], np.float32)]).astype(np.float32)  # [n, G, 3]
    cand_q = np.stack([q[[0, 1]], q[[2, 3]], q[[0, 1]]]).astype(np.float32)
    grasps = GoalToolPose.from_poses({frame: Pose(torch.as_tensor(cand_p.reshape(n * G, 3), device=device),
                                                  torch.as_tensor(cand_q.reshape(n * G, 4), device=device))}, num_goalset=G)
    assert grasps.position.shape == (n, 1, 1, G, 3)
    offset = -0.10
    g = planner.plan_grasp(grasps, cur, grasp_approach_offset=offset, grasp_lift_offset=offset)
    assert g.status == "Grasp planning completed."
    want = [True, True, False]
    for flags in (g.success, g.approach_success, g.grasp_success, g.lift_success):
        assert flags.cpu().tolist() == want, (g.status, flags)
    assert torch.equal(kp.link_spheres, kp.reference_link_spheres)
    contacts_off = dataclasses.replace(model, link_spheres=np.where(
        np.isin(model.link_sphere_idx_map, [model.link_names.index(x) for x in kp.grasp_contact_link_names if x in model.link_names])[None, :, None]
        & (np.arange(4) == 3)[None, None, :], np.float32(-100.0), model.link_spheres))
    for b in (0, 1):
        gi = int(g.goalset_index[b, 0])
        gp, gq = cand_p[b, gi], cand_q[b, gi]
        axis = _quat_rotate_np(gq, np.array([0.0, 0.0, 1.0], np.float32))
        legs = [(g.approach_trajectory, q0n[b], gp + offset * axis, model, False), (g.grasp_trajectory, None, gp, contacts_off, True),
                (g.lift_trajectory, None, gp + offset * axis, contacts_off, True)]
        prev = None
        for js, start, goal_p, m2, linear in legs:
            traj = js.position[b].cpu().numpy()  # [1, H, D]
            _verify_with_oracle(oracle, m2, arrays, traj, js.dt[b].cpu().numpy(), prev if start is None else start, rc)
            pp, qq = _tool_pose(oracle, model, traj[0])
            np.testing.assert_allclose(pp[-1], goal_p, atol=5e-3)
            if linear:
                d = pp - gp
                assert np.linalg.norm(d - (d @ axis)[:, None] * axis, axis=1).max() < 0.01
                assert np.minimum(np.linalg.norm(qq - gq, axis=1), np.linalg.norm(qq + gq, axis=1)).max() < 0.05
            prev = traj[0, -1]
    # approach only
    g2 = planner.plan_grasp(grasps, cur, grasp_approach_offset=offset, plan_approach_to_grasp=False)
    assert g2.success.cpu().tolist() == want and g2.grasp_trajectory is None and g2.status == "Planning to approach pose completed."


def test_scene_objects_reach_the_kernels(oracle, device, this_repos_curobo):  # noqa: F811
    """a world built from ``curobo.scene`` objects (cuboid + sphere + an ESDF ``VoxelGrid``) gives the collision costs of the
    oracle on the same obstacles, through ``scene_from_config`` and through a planner's ``update_world``"""
    from curobo.scene import Cuboid, Scene, Sphere, VoxelGrid, scene_from_config
    from curobo_amd.backends import collision as collision_hip
    from curobo_amd.scene import cuboid_scene_arrays, voxel_grid_from_sdf

    model = load_model("franka")
    md = model.as_dict()
    box = {"dims": [0.3, 0.3, 0.3], "pose": [0.5, 0.0, 0.4, 1, 0, 0, 0]}
    ball = {"type": "sphere", "radius": 0.15, "pose": [0.2, 0.4, 0.5, 1, 0, 0, 0]}
    centre, half = np.array([0.3, -0.3, 0.4]), np.array([0.1, 0.15, 0.2])

    def sdf(p):  # positive inside a box (the ESDF convention of the voxel store)
        d = np.abs(p - centre) - half
        return -(np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(1), 0))

    n, vs = 32, 0.025
    grid_arrays = voxel_grid_from_sdf(sdf, (n, n, n), vs, (0.3, -0.3, 0.4, 1, 0, 0, 0))
    esdf = torch.as_tensor(grid_arrays["voxel_features"].reshape(n, n, n).astype(np.float32))
    scene = Scene(cuboid=[Cuboid(name="box", **box)], sphere=[Sphere(name="ball", radius=0.15, pose=ball["pose"])],
                  voxel=[VoxelGrid(name="esdf", dims=[n * vs] * 3, voxel_size=vs, feature_tensor=esdf, pose=[0.3, -0.3, 0.4, 1, 0, 0, 0])])
    data = scene_from_config(scene, device)
    arrays = dict(cuboid_scene_arrays([[box, ball]]), **grid_arrays)
    q = sample_q(model, 64, seed=2, scale=0.6)
    sph = oracle.kinematics_forward(q, md)["robot_spheres"].reshape(64, 1, -1, 4)
    want = oracle.scene_collision(sph, arrays, 1.0, 0.02)
    S = sph.shape[2]
    dist, grad = torch.zeros(64, 1, S, device=device), torch.zeros(64, 1, S, 4, device=device)
    w, eta = torch.tensor([1.0], device=device), torch.tensor([0.02], device=device)
    env = torch.zeros(64, dtype=torch.int32, device=device)
    collision_hip.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), data.struct, w, eta, env, 64, 1, S, False, 0, False, w)
    torch.cuda.synchronize()
    assert (want["distance"] > 0).mean() > 0.02
    np.testing.assert_allclose(dist.cpu().numpy(), want["distance"], rtol=1e-4, atol=1e-5)
    # the same description through a planner
    from curobo.motion_planner import MotionPlanner, MotionPlannerCfg

    planner = MotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=None))
    assert planner.trajopt_solver.config.scene is None
    planner.update_world(scene)
    st = planner.trajopt_solver.config.scene.struct
    assert st.max_cuboids == 2 and st.max_voxel_grids == 1
