"""Model-predictive control loop (reference solver/solver_mpc.py): closed-loop tracking of a
tool-pose goal with warm-started re-optimisation, and re-planning when the goal moves."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("continuous", [False, True])
def test_mpc_tracks_a_pose_goal_in_closed_loop(continuous, oracle, device):
    from curobo_amd.kinematics import Kinematics, KinematicsCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import MPCSolver, MPCSolverCfg
    from curobo_amd.types import JointState
    from curobo_amd.workloads import c1_world, start_configuration

    kcfg = KinematicsCfg.from_packaged("franka", device=device)
    kin = kcfg.kinematics_config
    arrays = cuboid_scene_arrays(c1_world())
    scene = SceneData.from_arrays(arrays, device)
    B = 2
    fk = Kinematics(kcfg, compute_spheres=True)
    q0 = torch.as_tensor(start_configuration(kcfg.model), device=device).repeat(B, 1)
    # goals = FK of collision-free configurations near the start (reachable without crossing an obstacle)
    dq = torch.tensor([[0.5, 0.2, -0.3, 0.3, 0.2, -0.2, 0.3], [-0.5, 0.1, 0.3, 0.2, -0.3, 0.3, -0.2]], device=device)
    goal = fk.compute_kinematics(JointState.from_position((q0 + dq).unsqueeze(1))).tool_poses.as_goal()
    mpc = MPCSolver(kin, scene, B, MPCSolverCfg(continuous_commands=continuous))
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0))
    with pytest.raises(RuntimeError, match="setup"):
        mpc.optimize_next_action(state)
    mpc.setup(state, goal)
    times, reopt, errs = [], 0, []
    vmax = float(kin.joint_limits_velocity[1].abs().max())
    per_plan = 8 if continuous else 4
    for step in range(400):
        res = mpc.optimize_next_action(state)
        reopt += int(res.reoptimized)
        times.append((res.reoptimized, res.solve_time))
        a = res.next_action
        # continuity of the command stream: the next command is close to the current state (command_dt * max velocity)
        # continuity of the command stream: one command step (5 ms) of motion; the reference's scheme additionally steps by
        # about velocity x optimization_dt when a plan is renewed, the continuous scheme must not
        step_bound = vmax * mpc.command_dt * 1.1 + 1e-4 + (0.0 if continuous or not res.reoptimized else 0.06)
        assert float((a.position - state.position).abs().max()) < step_bound
        state = JointState(position=a.position, velocity=a.velocity, acceleration=a.acceleration)  # perfect tracking
        st = fk.compute_kinematics(JointState.from_position(state.position.unsqueeze(1)))
        errs.append((st.tool_poses.position[:, 0, 0] - goal.position[:, 0, 0, 0]).norm(dim=-1).max().item())
        assert bool(res.feasible.all())
    assert reopt == 400 // per_plan, reopt  # one re-optimisation per knot interval (two in the continuous scheme)
    assert errs[-1] < 0.005 and errs[-1] < 0.05 * errs[0], (errs[0], errs[-1])
    # the robot never touched the world on the way (oracle check of the executed states is done on the last one)
    sph = st.robot_spheres.cpu().numpy()
    assert float(oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum()) == 0.0
    cold = [t for r, t in times[:1]][0]
    warm = np.median([t for r, t in times[1:] if r])
    assert warm < cold, (warm, cold)
    # moving goal: swap the two robots' goals, the loop re-plans and converges again
    from curobo_amd.types import GoalToolPose

    goal2 = GoalToolPose(goal.tool_frames, goal.position.flip(0).contiguous(), goal.quaternion.flip(0).contiguous())
    mpc.update_goal_tool_poses(goal2)
    for step in range(1000):
        a = mpc.optimize_next_action(state).next_action
        state = JointState(position=a.position, velocity=a.velocity, acceleration=a.acceleration)
    st = fk.compute_kinematics(JointState.from_position(state.position.unsqueeze(1)))
    assert float((st.tool_poses.position[:, 0, 0] - goal2.position[:, 0, 0, 0]).norm(dim=-1).max()) < 0.01
    seq = mpc.optimize_action_sequence(state)
    assert seq.action_sequence.position.shape[0] == B and seq.action_buffer.shape == (B, 16, 7)


def test_model_predictive_control_front_end(oracle, device):
    """the reference's usage (curobo/model_predictive_control.py docstring): ModelPredictiveControlCfg.create(robot=..., scene_model=...) ->
    ModelPredictiveControl -> setup(current_state) -> update_goal_tool_poses -> optimize_next_action in a loop; per-robot goal updates
    through ``robot_ids``; a goal given as ``{tool frame: Pose}``"""
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg
    from curobo_amd.scene.types import Cuboid, SceneCfg
    from curobo_amd.types import JointState, Pose

    scene = SceneCfg(cuboid=[Cuboid(name="table", dims=[2.0, 2.0, 0.2], pose=[0, 0, -0.1, 1, 0, 0, 0])])
    B = 2
    config = ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=scene, optimization_dt=0.02, interpolation_steps=4,
                                              max_batch_size=B, optimizer_configs=["mpc/lbfgs_mpc.yml"])  # (accepted, ignored)
    mpc = ModelPredictiveControl(config)
    assert mpc.action_dim == 7 and mpc.action_horizon == 16 and abs(mpc.command_dt - 0.005) < 1e-12 and mpc.problem_batch_size == B
    q0 = mpc.default_joint_state.position.view(1, -1).repeat(B, 1)
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0), joint_names=mpc.joint_names)
    with pytest.raises(RuntimeError, match="setup"):
        mpc.optimize_next_action(state)
    mpc.setup(state)  # holds the current tool pose
    hold = mpc.optimize_next_action(state)
    assert float((hold.next_action.position - q0).abs().max()) < 2e-3
    dq = torch.tensor([[0.4, 0.2, -0.3, 0.3, 0.2, -0.2, 0.3], [-0.4, 0.1, 0.3, 0.2, -0.3, 0.3, -0.2]], device=device)
    goal = mpc.compute_kinematics(JointState.from_position(q0 + dq)).tool_poses.as_goal()
    assert mpc.update_goal_tool_poses(goal)  # (run_ik = True by default: the goal's IK solution is tracked in joint space as well)
    sol = mpc.solver
    assert float(sol.rollout._cs_tw) == 1000.0 and float(sol.metrics_rollout._cs_tw) == 1000.0
    reached = mpc.compute_kinematics(JointState.from_position(sol._goal_config)).tool_poses.position[:, 0, 0]
    assert float((reached - goal.position[:, 0, 0, 0]).norm(dim=-1).max()) < 5e-3, "the tracked configuration reaches the goal pose"
    assert mpc.update_goal_tool_poses(goal, run_ik=False) and float(sol.rollout._cs_tw) == 0.0, "poses alone"
    far = goal.clone()
    far.position[..., 0] += 3.0  # unreachable: the IK fails, the previous goal stays
    assert not mpc.update_goal_tool_poses(far) and float((mpc._goal.position - goal.position).abs().max()) == 0.0
    assert mpc.update_goal_tool_poses(goal)

    def run(steps):
        nonlocal state
        for _ in range(steps):
            r = mpc.optimize_next_action(state)
            state = JointState(position=r.next_action.position.clone(), velocity=r.next_action.velocity.clone(),
                               acceleration=r.next_action.acceleration.clone(), joint_names=mpc.joint_names)
        st = mpc.compute_kinematics(JointState.from_position(state.position))
        return st.tool_poses.position[:, 0, 0]

    p = run(400)
    assert float((p - goal.position[:, 0, 0, 0]).norm(dim=-1).max()) < 0.01
    # robot 1 alone gets a new goal (robot_ids); robot 0 keeps its own
    back = mpc.compute_kinematics(JointState.from_position(q0)).tool_poses.as_goal()
    mpc.update_goal_tool_poses(back, robot_ids=torch.tensor([1], device=device))
    p = run(400)
    assert float((p[0] - goal.position[0, 0, 0, 0]).norm()) < 0.01 and float((p[1] - back.position[1, 0, 0, 0]).norm()) < 0.01
    # a dictionary of poses: both robots to the start pose again; the plans stay clear of the table (oracle)
    frame = mpc.tool_frames[0]
    mpc.update_goal_tool_poses({frame: Pose(back.position[:, 0, 0, 0], back.quaternion[:, 0, 0, 0])})
    p = run(400)
    assert float((p - back.position[:, 0, 0, 0]).norm(dim=-1).max()) < 0.01
    model = config.kinematics.model
    sph = oracle.kinematics_forward(state.position.cpu().numpy(), model.as_dict())["robot_spheres"].reshape(B, 1, -1, 4)
    from curobo_amd.scene import cuboid_scene_arrays
    arrays = cuboid_scene_arrays([[{"dims": [2.0, 2.0, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}]])
    assert (oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"] == 0).all()
    seq = mpc.optimize_action_sequence(state)
    assert seq.action_sequence.position.shape[0] == B and seq.action_sequence.position.shape[-1] == 7


def test_mpc_joint_space_control(oracle, device):
    """``update_goal_state`` + ``enable_joint_position_tracking`` with the pose tracking switched off (reference solver_mpc.py:458-474,
    solver_core.py:392-414): the controller drives the robot to a goal CONFIGURATION"""
    from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg
    from curobo_amd.types import JointState

    mpc = ModelPredictiveControl(ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=None))
    q0 = mpc.default_joint_state.position.view(1, -1).clone()
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0), joint_names=mpc.joint_names)
    mpc.setup(state)
    q_goal = q0 + torch.tensor([[0.5, -0.3, 0.4, 0.3, -0.4, 0.3, 0.5]], device=q0.device)
    mpc.disable_tool_pose_tracking()
    mpc.update_goal_state(JointState.from_position(q_goal, mpc.joint_names))
    mpc.enable_joint_position_tracking()
    for _ in range(500):
        r = mpc.optimize_next_action(state)
        state = JointState(position=r.next_action.position.clone(), velocity=r.next_action.velocity.clone(),
                           acceleration=r.next_action.acceleration.clone(), joint_names=mpc.joint_names)
    err = (state.position - q_goal).abs().max()
    assert float(err) < 0.02, f"joint-space goal missed by {float(err):.3f} rad"
    lo, hi = mpc.kinematics.kinematics_config.joint_limits_position
    assert bool(((state.position >= lo) & (state.position <= hi)).all())
    with pytest.raises(ValueError, match="robot_ids"):
        mpc.update_goal_state(JointState.from_position(q_goal, mpc.joint_names), robot_ids=torch.tensor([0]))


def test_mpc_seed_trajectory_from_outside(oracle, device):
    """``update_seed_trajectory`` / ``update_seed_trajectory_from_goal_state`` (reference solver_mpc.py:498-531): the next solve starts
    from the given knots, a warm controller re-optimises at once, and a seed that
    already leads to the goal's configuration converges in the first solve where the hold-still seed does not"""
    from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg
    from curobo_amd.types import JointState

    mpc = ModelPredictiveControl(ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=None))
    q0 = mpc.default_joint_state.position.view(1, -1).clone()
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0), joint_names=mpc.joint_names)
    q_goal = q0 + torch.tensor([[0.4, -0.3, 0.3, 0.3, -0.3, 0.3, 0.4]], device=q0.device)
    goal = mpc.compute_kinematics(JointState.from_position(q_goal, mpc.joint_names)).tool_poses.as_goal()
    mpc.setup(state, goal)
    s = mpc.solver
    nk, D = s.rollout_cfg.n_knots, 7
    # shapes are checked as the reference checks them
    for bad in (torch.zeros(nk, D), torch.zeros(2, nk, D), torch.zeros(1, nk + 1, D), torch.zeros(1, nk, D + 1)):
        with pytest.raises(ValueError, match="seed_trajectory"):
            mpc.update_seed_trajectory(bad.to(device))
    # the solve starts from the seed (limits applied), once
    started_from, solve = [], s._solve
    s._solve = lambda knots, iters: (started_from.append(knots.clone()), solve(knots, iters))[1]
    seed = (q0.view(1, 1, D) + torch.linspace(0, 1, nk, device=device).view(1, nk, 1) * 0.2).contiguous()
    mpc.update_seed_trajectory(seed)
    r = mpc.optimize_next_action(state)
    assert r.reoptimized and torch.equal(started_from[-1], seed) and s._seed_override is None
    # a warm controller with commands left re-optimises on the next call once a seed is handed in
    r = mpc.optimize_next_action(state)
    assert not r.reoptimized
    mpc.update_seed_trajectory_from_goal_state(JointState.from_position(q_goal, mpc.joint_names))
    want = q0.view(1, 1, D) + torch.linspace(0, 1, nk + 2, device=device)[1:-1].view(1, nk, 1) * (q_goal - q0).view(1, 1, D)
    assert torch.allclose(s._seed_override, want, atol=1e-6)
    r = mpc.optimize_next_action(state)
    assert r.reoptimized and s._seed_override is None and torch.allclose(started_from[-1], want, atol=1e-6)
    # the line to the goal configuration as the first seed: the plan's last point is nearer the goal after ONE cold solve than from the hold-still seed
    mpc.reset_robot(state)
    mpc.update_seed_trajectory_from_goal_state(JointState.from_position(q_goal, mpc.joint_names))
    r_seeded = mpc.optimize_action_sequence(state)
    mpc.reset_robot(state)
    r_plain = mpc.optimize_action_sequence(state)
    assert float(r_seeded.position_error.max()) < 0.05, (r_seeded.position_error, r_plain.position_error)
    assert float(r_seeded.position_error.max()) <= float(r_plain.position_error.max()) + 1e-4


def test_mpc_around_a_mesh_obstacle(oracle, device):
    """the front end over a scene description with a triangle mesh: a ball the hand's arc grazes (8 cm deep on the straight
    joint-space line).  The loop re-optimises from captured graphs every knot interval -- hundreds of replays of graphs that hold
    the queued mesh launch -- reaches the goal, and every executed state is clear of table and ball by the oracle's brute force."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from oracle.oracle import mesh_scene_arrays
    from test_oracle_mesh import sphere_shape

    from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.types import JointState

    vs, fs = sphere_shape(0.08)
    table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
    ball = {"vertices": vs, "faces": fs, "pose": [0.6968, -0.1375, 0.3710, 1, 0, 0, 0]}
    config = ModelPredictiveControlCfg.create(robot="franka.yml", scene_model={"cuboid": {"table": table}, "mesh": {"ball": ball}},
                                              optimization_dt=0.02, interpolation_steps=4, max_batch_size=1)
    mpc = ModelPredictiveControl(config)
    assert config.scene.meshes is not None and config.scene.meshes.gradient_mode == config.scene.meshes.CONSISTENT_GRADIENT
    model = config.kinematics.model
    arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[dict(ball, name="ball")]])}
    q0 = torch.tensor([[-0.6, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]], device=device)
    q1 = q0.clone()
    q1[0, 0] = 0.4
    # the straight line between the two is in collision with the ball, its ends are not
    tt = np.linspace(0, 1, 41, dtype=np.float32)[:, None]
    line = q0.cpu().numpy() * (1 - tt) + q1.cpu().numpy() * tt
    pen = oracle.scene_collision(oracle.kinematics_forward(line, model.as_dict(), horizon=41)["robot_spheres"].reshape(1, 41, -1, 4),
                                 arrays, 1.0, 0.0)["distance"][0].sum(-1)
    assert pen[0] == 0 and pen[-1] == 0 and pen.max() > 0.05
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0), joint_names=mpc.joint_names)
    mpc.setup(state)
    goal = mpc.compute_kinematics(JointState.from_position(q1)).tool_poses.as_goal()
    assert mpc.update_goal_tool_poses(goal)
    visited = []
    for _ in range(700):
        r = mpc.optimize_next_action(state)
        state = JointState(position=r.next_action.position.clone(), velocity=r.next_action.velocity.clone(),
                           acceleration=r.next_action.acceleration.clone(), joint_names=mpc.joint_names)
        visited.append(state.position[0].cpu().numpy())
    p = mpc.compute_kinematics(JointState.from_position(state.position)).tool_poses.position[:, 0, 0]
    assert float((p - goal.position[:, 0, 0, 0]).norm(dim=-1).max()) < 0.015
    v = np.stack(visited[::4])
    sph = oracle.kinematics_forward(v, model.as_dict())["robot_spheres"].reshape(len(v), 1, -1, 4)
    d = oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum((1, 2))
    assert (d == 0).all(), (int((d > 0).sum()), float(d.max()))
