"""Goals for EVERY tool frame through the IK solvers and the planners (reference: the whole GoalToolPose goes to
``ik_solver.solve_pose``, motion/motion_planner.py:249; success = converged on all frames, solver_ik.py:463-476).
Robots: dual_ur10e (two arms, T = 2) and the Unitree G1 (hands + feet, T = 4, BASELINE config 4's robot).  Every
solution is verified with the oracle's FK, frame by frame."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q

pytestmark = pytest.mark.gpu


def _free_configs(oracle, model, n, seed, scale):
    """n configurations without self collision (oracle), and their tool poses [n, T, 3 | 4]"""
    cand = sample_q(model, 40 * n, seed=seed, scale=scale)
    fk = oracle.kinematics_forward(cand, model.as_dict())
    sph = fk["robot_spheres"].reshape(len(cand), 1, -1, 4)
    free = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"].reshape(-1) == 0
    sel = np.nonzero(free)[0][:n]
    assert len(sel) == n, f"only {len(sel)} of {len(cand)} sampled configurations are free of self collision"
    return cand[sel], fk["link_pos"][sel], fk["link_quat"][sel]


def _frame_errors(oracle, model, q, gp, gq):
    """position / rotation error of every tool frame of configurations q [n, D] against goals [n, T, 3 | 4]"""
    fk = oracle.kinematics_forward(np.ascontiguousarray(q, np.float32), model.as_dict())
    pe = np.linalg.norm(fk["link_pos"] - gp, axis=-1)
    dot = np.abs((fk["link_quat"] * gq).sum(-1))
    return pe, 2 * np.arccos(np.clip(dot, 0, 1))


def test_ik_solver_reaches_the_goal_of_every_tool_frame_dual_arm(oracle, device):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.solver import IKSolver, IKSolverCfg

    model = load_model("dual_ur10e")
    kin = KinematicsParams.from_model(model, device)
    assert kin.num_pose_links == 2 and list(kin.tool_frames) == list(model.tool_frames)
    P = 8
    q_goal, gp, gq = _free_configs(oracle, model, P, seed=5, scale=0.6)
    solver = IKSolver(kin, None, P, IKSolverCfg(num_seeds=32))
    res = solver.solve_pose(torch.as_tensor(gp).view(P, 2, 1, 3), torch.as_tensor(gq).view(P, 2, 1, 4), exit_early=False)
    torch.cuda.synchronize()
    ok = res.success.cpu().numpy()
    assert ok.mean() >= 0.75, f"two-frame IK success rate {ok.mean():.2f}"
    assert res.goalset_index.shape == (P, 2)
    pe, re = _frame_errors(oracle, model, res.solution.cpu().numpy(), gp, gq)
    assert (pe[ok] < 5e-3).all() and (re[ok] < 0.05).all(), "a success holds on BOTH frames"
    # the reported errors are the largest over the frames
    np.testing.assert_allclose(res.position_error.cpu().numpy()[ok], pe[ok].max(-1), atol=2e-4)
    # a goal only ONE arm can reach is not a success: arm 2's goal 3 m away
    far = gp.copy()
    far[:, 1] += np.array([3.0, 0.0, 0.0], np.float32)
    bad = solver.solve_pose(torch.as_tensor(far).view(P, 2, 1, 3), torch.as_tensor(gq).view(P, 2, 1, 4), exit_early=False)
    assert not bool(bad.success.any()) and float(bad.position_error.min()) > 0.5
    # the single-frame call forms are for single-frame robots
    with pytest.raises(ValueError, match="tool frames"):
        solver.solve_pose(torch.as_tensor(gp[:, 0]), torch.as_tensor(gq[:, 0]))
    # top-k
    top = solver.solve_pose(torch.as_tensor(gp).view(P, 2, 1, 3), torch.as_tensor(gq).view(P, 2, 1, 4), return_seeds=3, exit_early=False)
    assert top.solution.shape == (P, 3, kin.num_dof) and top.goalset_index.shape == (P, 3, 2)
    pe3, re3 = _frame_errors(oracle, model, top.solution[:, 0].cpu().numpy(), gp, gq)
    ok3 = top.success[:, 0].cpu().numpy()
    assert ok3.mean() >= 0.75 and (pe3[ok3] < 5e-3).all() and (re3[ok3] < 0.05).all()


def test_ik_solver_four_tool_frames_humanoid(oracle, device):
    """Unitree G1: both hands and both feet at once (T = 4, 49 dof)"""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.solver import IKSolver, IKSolverCfg

    model = load_model("unitree_g1")
    kin = KinematicsParams.from_model(model, device)
    assert kin.num_pose_links == 4
    P = 2
    q_goal, gp, gq = _free_configs(oracle, model, P, seed=3, scale=0.25)
    solver = IKSolver(kin, None, P, IKSolverCfg(num_seeds=16))
    res = solver.solve_pose(torch.as_tensor(gp).view(P, 4, 1, 3), torch.as_tensor(gq).view(P, 4, 1, 4), exit_early=False)
    torch.cuda.synchronize()
    ok = res.success.cpu().numpy()
    pe, re = _frame_errors(oracle, model, res.solution.cpu().numpy(), gp, gq)
    print(f"\n[g1 ik] success {ok.tolist()}, largest frame errors {pe.max(-1).tolist()} m / {re.max(-1).tolist()} rad")
    assert ok.any(), "at least one of the two whole-body problems is solved"
    assert (pe[ok] < 5e-3).all() and (re[ok] < 0.05).all(), "a success holds on all four frames"
    assert res.goalset_index.shape == (P, 4)


def test_front_ends_take_every_frame_in_any_order(oracle, device):
    """InverseKinematics.solve_pose and MotionPlanner.plan_pose on the dual arm: the GoalToolPose names both frames (here in
    the opposite order of the robot's), a goal with a frame missing is refused, and the plan ends in both poses."""
    from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
    from curobo_amd.types import GoalToolPose, JointState, Pose

    model = load_model("dual_ur10e")
    frames = list(model.tool_frames)
    B = 4
    q_goal, gp, gq = _free_configs(oracle, model, B, seed=9, scale=0.5)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    poses = {frames[i]: Pose(t(gp[:, i]), t(gq[:, i])) for i in range(2)}
    goal = GoalToolPose.from_poses(poses, ordered_tool_frames=frames[::-1])  # reversed order
    ik = InverseKinematics(InverseKinematicsCfg.create(robot="dual_ur10e.yml", scene_model=None, num_seeds=32))
    r = ik.solve_pose(goal)
    ok = r.success[:, 0].cpu().numpy()
    assert ok.mean() >= 0.75 and r.goalset_index.shape == (B, 1, 2)
    pe, re = _frame_errors(oracle, model, r.solution[:, 0].cpu().numpy(), gp, gq)
    assert (pe[ok] < 5e-3).all() and (re[ok] < 0.05).all()
    with pytest.raises(ValueError, match="every tool frame"):
        ik.solve_pose(GoalToolPose.from_poses({frames[0]: poses[frames[0]]}))
    # disable_tool_pose_tracking (reference solver_core.py:392-401): the second arm's goal 3 m away no longer counts ...
    far = gp.copy()
    far[:, 1] += np.array([3.0, 0.0, 0.0], np.float32)
    far_goal = GoalToolPose.from_poses({frames[i]: Pose(t(far[:, i]), t(gq[:, i])) for i in range(2)})
    assert not bool(ik.solve_pose(far_goal).success.any())
    ik.disable_tool_pose_tracking([frames[1]])
    r1 = ik.solve_pose(far_goal)
    ok1 = r1.success[:, 0].cpu().numpy()
    assert ok1.mean() >= 0.75
    pe1, _ = _frame_errors(oracle, model, r1.solution[:, 0].cpu().numpy(), far, gq)
    assert (pe1[ok1, 0] < 5e-3).all() and (pe1[ok1, 1] > 1.0).all(), "frame 0 reached, frame 1 ignored"
    # ... and enable_tool_pose_tracking brings it back
    ik.enable_tool_pose_tracking()
    assert not bool(ik.solve_pose(far_goal).success.any())
    assert ik.solve_pose(goal).success[:, 0].float().mean() >= 0.75
    # ---- the planner: IK over both frames -> trajectory optimisation over both frames
    planner = MotionPlanner(MotionPlannerCfg.create(robot="dual_ur10e.yml", scene_model=None, num_ik_seeds=32, num_trajopt_seeds=4))
    cur = JointState.from_position(planner.default_joint_state.position.view(1, -1).clone(), planner.joint_names)
    q0 = cur.position[0].cpu().numpy()
    near = (q0[None] + np.array([[0.4, -0.3, 0.3, 0.2, 0.1, 0.0, -0.3, 0.2, -0.3, 0.1, 0.2, 0.0]], np.float32)).astype(np.float32)
    fk = oracle.kinematics_forward(near, model.as_dict())
    one = GoalToolPose.from_poses({frames[i]: Pose(t(fk["link_pos"][:, i]), t(fk["link_quat"][:, i])) for i in range(2)},
                                  ordered_tool_frames=frames[::-1])
    res = planner.plan_pose(one, cur, max_attempts=3)
    assert res is not None and bool(res.success[0, 0]), "plan_pose reaches a two-frame goal"
    end = res.js_solution.position[0, 0, -1].cpu().numpy()[None]
    pe, re = _frame_errors(oracle, model, end, fk["link_pos"], fk["link_quat"])
    assert (pe < 5e-3).all() and (re < 0.05).all(), (pe, re)
