"""Time-optimal finetune passes of the trajectory optimiser, ``solve_cspace``, seeds from the caller, and the planner
front ends (``curobo.trajectory_optimizer`` / ``curobo.motion_planner`` / ``curobo.batch_motion_planner``): reference
``curobo/_src/solver/solver_trajopt.py:258-467,831-971``, ``motion/motion_planner.py:207-396``,
``motion/motion_planner_batch.py:139-289``.  Winning trajectories are verified with the oracle."""

import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_model

pytestmark = pytest.mark.gpu


def _setup(device, world="c2"):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c1_world, c2_world

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c2_world() if world == "c2" else c1_world())
    return model, kin, arrays, SceneData.from_arrays(arrays, device)


def _verify_with_oracle(oracle, model, arrays, traj, dt, start, rc, env=None):
    """traj [n, H, D] at dt [n]: starts at ``start``, inside the joint limits, free of self / scene collision, and its
    finite-difference velocity respects the limits"""
    n, H, D = traj.shape
    md = model.as_dict()
    np.testing.assert_allclose(traj[:, 0], np.broadcast_to(start, (n, D)), atol=1e-4)
    lo, hi = model.joint_limits_position
    assert (traj >= lo - 1e-3).all() and (traj <= hi + 1e-3).all()
    chk = oracle.kinematics_forward(traj.reshape(n * H, D), md, horizon=H)
    s2 = chk["robot_spheres"].reshape(n, H, -1, 4)
    assert (oracle.self_collision(s2, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all()
    kw = dict(env_query_idx=env.astype(np.int32), use_multi_env=True) if env is not None else {}
    assert (oracle.scene_collision(s2, arrays, 1.0, 0.0, **kw)["distance"].sum((1, 2)) == 0).all()
    vel = np.diff(traj, axis=1) / dt[:, None, None]
    vmax = np.abs(model.joint_limits_velocity).max(0)
    assert (np.abs(vel) <= vmax * 1.02 + 1e-3).all()
    return chk


def test_finetune_passes_shorten_the_motion_and_keep_it_feasible(oracle, device):
    """reference _solve_impl (:337-450): every pass re-solves at 0.55 x the best dt so far from the previous winner and a
    seed only takes the new solution when it succeeded at a dt that is not slower -> with finetune passes the winner is
    never slower, and on this workload clearly faster; the retimed winners pass the velocity / acceleration / jerk /
    collision checks recomputed with the oracle's B-spline at the reported dt."""
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import feasible_goals, start_configuration

    model, kin, arrays, scene = _setup(device)
    P = 8
    gp, gq = feasible_goals(kin, scene, P)
    start = torch.as_tensor(start_configuration(model))
    slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=4))
    r0 = slv.solve_pose(start, gp, gq, finetune_attempts=0)
    slv.reset_seed()  # (the Halton stream of the IK's seed stage runs on between solves: same seeds for both)
    r2 = slv.solve_pose(start, gp, gq, finetune_attempts=2)
    torch.cuda.synchronize()
    assert r0.finetune_passes == 1 and 2 <= r2.finetune_passes <= 3
    ok0, ok2 = r0.success.cpu().numpy(), r2.success.cpu().numpy()
    assert ok0.mean() >= 0.75 and ok2.mean() >= 0.75
    assert (ok2 | ~ok0).all(), "a finetune pass never loses a solved problem (it only replaces successful seeds)"
    both = ok0 & ok2
    t0, t2 = r0.motion_time.cpu().numpy(), r2.motion_time.cpu().numpy()
    assert (t2[both] <= t0[both] * (1 + 1e-5)).all()
    # seed by seed: a seed that was solved keeps its solution unless a later pass solved it at a dt that is not slower;
    # some seeds do get faster (the passes run at 0.55 x the dt and the result is retimed: what can change is the shape
    # of the velocity profile, the knots of a straight joint-space line are evenly spaced already)
    a0, a2 = r0.all_seeds, r2.all_seeds
    s_both = (a0["success"] & a2["success"]).cpu().numpy()
    d0, d2 = a0["traj_dt"].cpu().numpy(), a2["traj_dt"].cpu().numpy()
    assert (d2[s_both] <= d0[s_both] * (1 + 1e-6)).all() and (d2[s_both] < d0[s_both] * 0.995).any()
    # ... and every solved seed is much faster than its seed trajectory (the straight line at the fastest dt ITS velocity /
    # acceleration / jerk allow): the first pass already runs at 0.55 x that dt
    seed_dt = slv.last_pass_trace[0]["seed_dt"].cpu().numpy()
    assert np.median(d2[s_both] / seed_dt[s_both]) < 0.9, (d2, seed_dt)
    rc, cfg = slv.cfg.rollout, slv.cfg
    for r, ok in ((r0, ok0), (r2, ok2)):
        dt = r.traj_dt.cpu().numpy()
        assert (dt >= cfg.minimum_trajectory_dt - 1e-7).all() and (dt <= cfg.maximum_trajectory_dt + 1e-7).all()
        traj = r.position.cpu().numpy()[ok]
        _verify_with_oracle(oracle, model, arrays, traj, dt[ok], start.numpy(), rc)
        # the trajectory is the B-spline of the returned knots at the returned dt, ending at rest in goal_config
        D, H = kin.num_dof, rc.padded_horizon
        z = np.zeros((1, D), np.float32)
        st = {"position": start.numpy().reshape(1, D).astype(np.float32), "velocity": z, "acceleration": z, "jerk": z}
        zp = np.zeros((P, D), np.float32)
        gl = {"position": r.goal_config.cpu().numpy().astype(np.float32), "velocity": zp, "acceleration": zp, "jerk": zp}
        s = oracle.bspline_forward(r.knots.cpu().numpy(), st, gl, np.zeros(P, np.int32), np.arange(P, dtype=np.int32),
                                   dt.astype(np.float32), np.ones(P, np.uint8), H, rc.bspline_degree)
        np.testing.assert_allclose(s["position"][ok], traj, atol=1e-4)
        for name, lim in (("velocity", np.abs(model.joint_limits_velocity).max(0)), ("acceleration", rc.max_acceleration),
                          ("jerk", rc.max_jerk)):
            got = getattr(r, name).cpu().numpy()[ok]
            np.testing.assert_allclose(got, s[name][ok], rtol=2e-3, atol=2e-3 * np.abs(s[name][ok]).max())
            assert (np.abs(got) <= lim * (1 + 2e-3) + 1e-3).all(), name
        np.testing.assert_allclose(traj[:, -1], r.goal_config.cpu().numpy()[ok], atol=1e-4)
    # per-seed book-keeping: every successful seed's dt is inside the range, the winner is a successful seed with the
    # lowest rank cost
    a = r2.all_seeds
    assert a["success"].shape == (P, 4) and a["knots"].shape == (P, 4, rc.n_knots, kin.num_dof)
    best = a["cost"].argmin(1)
    assert torch.equal(best, r2.seed_index)


def test_solve_cspace_and_caller_seeds(oracle, device):
    """solve_cspace (reference :831-971): the goal is a joint configuration, the tool-pose target its forward kinematics,
    every trajectory ends exactly there.  seed_traj / seed_config / return_seeds / dt of the reference's solve_pose."""
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import start_configuration

    model, kin, arrays, scene = _setup(device, "c1")
    P, S = 4, 4
    goal_q = RobotCollisionChecker(KinematicsCfg(kin, None), scene).sample(P, mask_valid=True)
    assert goal_q.shape[0] == P
    start = torch.as_tensor(start_configuration(model))
    slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
    r = slv.solve_cspace(start, goal_q, finetune_attempts=1)
    torch.cuda.synchronize()
    ok = r.success.cpu().numpy()
    assert ok.mean() >= 0.75, ok
    traj, dt = r.position.cpu().numpy(), r.traj_dt.cpu().numpy()
    np.testing.assert_allclose(traj[:, -1], goal_q.cpu().numpy(), atol=1e-4)  # implicit goal state: exact, solved or not
    np.testing.assert_allclose(r.goal_config.cpu().numpy(), goal_q.cpu().numpy(), atol=1e-6)
    _verify_with_oracle(oracle, model, arrays, traj[ok], dt[ok], start.numpy(), slv.cfg.rollout)
    assert float(r.position_error[r.success].max()) < slv.cfg.position_threshold
    # the winners of that solve as seed trajectories of a pose solve (reference seed_traj [batch, n, n_knots, dof]; the
    # remaining seeds are lines to seed_config), two solutions per problem back
    gp, gq = slv._tool_pose_of(goal_q)
    seed_traj = r.knots.view(P, 1, slv.cfg.rollout.n_knots, kin.num_dof)
    seed_cfg = goal_q.view(P, 1, -1).expand(P, S, -1)
    r2 = slv.solve_pose(start, gp[:, 0], gq[:, 0], seed_config=seed_cfg, seed_traj=seed_traj, return_seeds=2, finetune_attempts=1)
    torch.cuda.synchronize()
    assert r2.success.shape == (P, 2) and r2.knots.shape == (P, 2, slv.cfg.rollout.n_knots, kin.num_dof) and r2.position.shape[:2] == (P, 2)
    assert (r2.cost[:, 0] <= r2.cost[:, 1]).all(), "returned seeds are ranked best first"
    assert r2.success[:, 0].float().mean() >= 0.75
    assert (r2.seed_index[:, 0] != r2.seed_index[:, 1]).all()
    # a given dt [batch, num_seeds] is where the first pass starts from (x finetune_dt_scale), not where it ends
    r3 = slv.solve_pose(start, gp[:, 0], gq[:, 0], seed_config=seed_cfg, dt=torch.full((P, S), 0.1), finetune_attempts=0)
    assert r3.finetune_passes == 1 and bool((r3.traj_dt > 0).all())
    # too few configurations for the seeds is the reference's error
    with pytest.raises(ValueError, match="Insufficient seed configs"):
        slv.solve_pose(start, gp[:, 0], gq[:, 0], seed_config=seed_cfg[:, :2])
    with pytest.raises(ValueError, match="seed_traj"):
        slv.solve_pose(start, gp[:, 0], gq[:, 0], seed_traj=seed_traj[:, :, :5])


@pytest.fixture
def this_repos_curobo():
    stale = [m for m in sys.modules if (m == "curobo" or m.startswith("curobo.")) and not str(getattr(sys.modules[m], "__file__", "")).startswith(ROOT)]
    saved = {m: sys.modules.pop(m) for m in stale}
    path = list(sys.path)
    sys.path[:] = [ROOT] + [p for p in sys.path if p != ROOT]
    yield
    sys.path[:] = path
    for m in [m for m in sys.modules if m == "curobo" or m.startswith("curobo.")]:
        sys.modules.pop(m)
    sys.modules.update(saved)


def _scene_cfg(world):
    return {"cuboid": {f"o{i}": {"dims": o["dims"], "pose": o["pose"]} for i, o in enumerate(world)}}


def test_motion_planner_plan_pose_and_plan_cspace(oracle, device, this_repos_curobo):
    """the reference's usage (curobo/motion_planner.py docstring): MotionPlannerCfg.create(robot=...) -> MotionPlanner ->
    plan_pose(goal_tool_poses, current_state) = IK -> trajectory optimisation -> time-optimal finetune -> interpolated plan"""
    from curobo.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo.types import JointState
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.workloads import c1_world

    world = c1_world()[0]
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=_scene_cfg(world), num_ik_seeds=32, num_trajopt_seeds=4,
                                     graph_planner_config="graph_planner/exact_graph_planner.yml")  # (accepted, ignored)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    arrays = cuboid_scene_arrays([world])
    cur = JointState.from_position(planner.default_joint_state.position.view(1, -1).clone(), planner.joint_names)
    goal_js = cur.clone()
    goal_js.position[0, 0] += 0.6
    goal_js.position[0, 3] += 0.3
    goal = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    res = planner.plan_pose(goal, cur)
    assert res is not None and bool(res.success[0, 0]), res
    assert res.js_solution.position.shape == (1, 1, 33, 7) and res.js_solution.dt.shape == (1, 1)
    assert float(res.position_error[0, 0]) < 0.005 and float(res.rotation_error[0, 0]) < 0.05
    traj = res.js_solution.position[0].cpu().numpy()
    chk = _verify_with_oracle(oracle, model, arrays, traj, res.js_solution.dt[0].cpu().numpy(), cur.position[0].cpu().numpy(),
                              config.trajopt_solver_config.solver_cfg().rollout)
    np.testing.assert_allclose(chk["link_pos"].reshape(1, 33, 3)[0, -1], goal.position[0, 0, 0, 0].cpu().numpy(), atol=5e-3)
    # the interpolated plan: interpolation_dt samples of the same spline, trimmed to the last step
    plan = res.get_interpolated_plan()
    n = plan.position.shape[0]
    assert n == int(res.interpolated_last_tstep[0, 0]) and n > 10
    np.testing.assert_allclose(plan.position[0].cpu().numpy(), traj[0, 0], atol=1e-4)
    np.testing.assert_allclose(plan.position[-1].cpu().numpy(), traj[0, -1], atol=2e-3)
    # every knot interval is rounded UP to whole interpolation steps (reference calculate_traj_steps, nearest_int): the
    # interpolated plan takes at least the optimised motion time and at most one step per knot interval longer
    extra = (n - 1) * 0.025 - float(res.motion_time[0, 0])
    assert -1e-4 <= extra <= 17 * 0.025
    assert res.total_time >= res.solve_time > 0.0
    # joint-space goal
    res_c = planner.plan_cspace(goal_js, cur)
    assert res_c is not None and bool(res_c.success[0, 0])
    np.testing.assert_allclose(res_c.js_solution.position[0, 0, -1].cpu().numpy(), goal_js.position[0].cpu().numpy(), atol=1e-4)
    # an unreachable goal: IK never succeeds -> None, as in the reference
    far = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    far.position[..., 2] += 3.0
    assert planner.plan_pose(far, cur, max_attempts=1) is None


def test_motion_planner_in_a_mesh_world(oracle, device, this_repos_curobo):
    """the same planner call with triangle-mesh obstacles in ``scene_model`` (reference SceneCfg ``mesh`` entries, geom/types.py):
    the mesh launch (cell lists + BVH walk) runs inside the captured solver and metrics graphs next to the cuboid store; the
    winner is free of collision against cuboids AND meshes by the oracle's brute force over every triangle.  The scenes the
    solvers build use the sign-consistent mesh gradient (``scene_from_config``): with the vector the reference's mesh query
    returns (data_mesh.py:693-697, kept as mode 0 of the launch) no seed of this problem leaves the pillar
    (tools/r06/mesh_vs_cuboid_plan.py, docs/NOTEBOOK.md round 6)."""
    from curobo.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo.types import JointState
    from oracle.oracle import mesh_scene_arrays
    from test_oracle_mesh import box_shape, sphere_shape

    from curobo_amd.scene import cuboid_scene_arrays

    vb, fb = box_shape([0.16, 0.16, 0.7], 2)
    vs, fs = sphere_shape(0.12)
    meshes = {"pillar": {"vertices": vb, "faces": fb, "pose": [0.5, 0.0, 0.35, 1, 0, 0, 0]},
              "ball": {"vertices": vs, "faces": fs, "pose": [0.0, 0.55, 0.9, 0.9238795, 0, 0.3826834, 0]}}
    table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model={"cuboid": {"table": table}, "mesh": meshes},
                                     num_ik_seeds=32, num_trajopt_seeds=4)
    planner = MotionPlanner(config)
    scene = config.trajopt_solver_config.scene
    assert scene.meshes is not None and len(scene.meshes.meshes) == 2 and all(m.cell_start is not None for m in scene.meshes.meshes)
    assert scene.meshes.gradient_mode == scene.meshes.CONSISTENT_GRADIENT
    model = config.trajopt_solver_config.kinematics.model
    arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[dict(m, name=k) for k, m in meshes.items()]])}
    q0 = torch.tensor([[-0.9, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]], device=planner.default_joint_state.position.device)
    cur = JointState.from_position(q0, planner.joint_names)
    goal_js = cur.clone()
    goal_js.position[0, 0] = 0.9  # the straight joint-space line sweeps the outstretched arm through the pillar
    H = 33
    tt = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    line = cur.position[0].cpu().numpy()[None] * (1 - tt) + goal_js.position[0].cpu().numpy()[None] * tt
    s_line = oracle.kinematics_forward(line, model.as_dict(), horizon=H)["robot_spheres"].reshape(1, H, -1, 4)
    assert oracle.scene_collision(s_line, arrays, 1.0, 0.0)["distance"].sum() > 0
    goal = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    res = planner.plan_pose(goal, cur, max_attempts=3)
    assert res is not None and bool(res.success[0, 0]), res
    assert float(res.position_error[0, 0]) < 0.005 and float(res.rotation_error[0, 0]) < 0.05
    traj = res.js_solution.position[0].cpu().numpy()
    _verify_with_oracle(oracle, model, arrays, traj, res.js_solution.dt[0].cpu().numpy(), cur.position[0].cpu().numpy(),
                        config.trajopt_solver_config.solver_cfg().rollout)
    # the pillar lifted out of the way IN PLACE (the captured graphs read the pose buffer): the next plan is shorter; put back,
    # the plan goes around it again
    around = float(res.motion_time[0, 0])
    scene.update_obstacle_pose("pillar", [0.5, 0.0, 3.35, 1, 0, 0, 0])
    res_free = planner.plan_pose(goal, cur, max_attempts=3)
    assert res_free is not None and bool(res_free.success[0, 0]) and float(res_free.motion_time[0, 0]) < 0.85 * around
    scene.update_obstacle_pose("pillar", [0.5, 0.0, 0.35, 1, 0, 0, 0])
    res_back = planner.plan_pose(goal, cur, max_attempts=3)
    assert res_back is not None and bool(res_back.success[0, 0])
    _verify_with_oracle(oracle, model, arrays, res_back.js_solution.position[0].cpu().numpy(), res_back.js_solution.dt[0].cpu().numpy(),
                        cur.position[0].cpu().numpy(), config.trajopt_solver_config.solver_cfg().rollout)


def test_motion_planner_in_a_voxel_world(oracle, device, this_repos_curobo):
    """the same problem with the pillar as an ESDF voxel grid (reference ``VoxelGrid`` entry of a scene description: dims,
    voxel_size, feature_tensor, pose): the planner's graphs run the voxel lookup next to the cuboid store; the winner is clear
    of table and grid by the oracle's voxel restatement"""
    from curobo.motion_planner import MotionPlanner, MotionPlannerCfg
    from curobo.types import JointState

    from curobo_amd.scene import cuboid_scene_arrays, voxel_grid_from_sdf
    from curobo_amd.scene.config import voxel_arrays_from_config

    centre, half = np.array([0.5, 0.0, 0.35]), np.array([0.08, 0.08, 0.35])

    def pillar(p):
        q = np.abs(p - centre) - half
        return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)

    pose = [0.5, 0.0, 0.45, 1, 0, 0, 0]
    grid = voxel_grid_from_sdf(pillar, (32, 32, 48), 0.02, pose7=pose, max_distance=10.0)
    table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
    world = {"cuboid": {"table": table}, "voxel": {"pillar": {"dims": [0.64, 0.64, 0.96], "voxel_size": 0.02, "pose": pose,
                                                              "feature_tensor": grid["voxel_features"].reshape(-1)}}}
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=world, num_ik_seeds=32, num_trajopt_seeds=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    arrays = {**cuboid_scene_arrays([[table]]), **voxel_arrays_from_config(world)}
    q0 = torch.tensor([[-0.9, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]], device=planner.default_joint_state.position.device)
    cur = JointState.from_position(q0, planner.joint_names)
    goal_js = cur.clone()
    goal_js.position[0, 0] = 0.9
    H = 33
    tt = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    line = cur.position[0].cpu().numpy()[None] * (1 - tt) + goal_js.position[0].cpu().numpy()[None] * tt
    s_line = oracle.kinematics_forward(line, model.as_dict(), horizon=H)["robot_spheres"].reshape(1, H, -1, 4)
    assert oracle.scene_collision(s_line, arrays, 1.0, 0.0)["distance"].sum() > 0.5
    goal = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    res = planner.plan_pose(goal, cur, max_attempts=3)
    assert res is not None and bool(res.success[0, 0]), res
    assert float(res.position_error[0, 0]) < 0.005 and float(res.rotation_error[0, 0]) < 0.05
    traj = res.js_solution.position[0].cpu().numpy()
    _verify_with_oracle(oracle, model, arrays, traj, res.js_solution.dt[0].cpu().numpy(), cur.position[0].cpu().numpy(),
                        config.trajopt_solver_config.solver_cfg().rollout)


def test_batch_motion_planner_one_world_per_problem(oracle, device, this_repos_curobo):
    """BASELINE config 5 at planner level (reference motion_planner_batch.py with multi_env): a batch of problems, each with
    its own start state, goal and world; every winner is collision free in ITS world."""
    from curobo.batch_motion_planner import BatchMotionPlanner, MotionPlannerCfg
    from curobo.types import JointState
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.workloads import c1_world, c2_world

    B = 4
    worlds = [c1_world()[0], c2_world()[0], c1_world()[0], c2_world()[0]]
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=[_scene_cfg(w) for w in worlds], max_batch_size=B,
                                     multi_env=True, num_ik_seeds=32, num_trajopt_seeds=4)
    planner = BatchMotionPlanner(config)
    assert planner.batch_size == B
    model = config.trajopt_solver_config.kinematics.model
    arrays = cuboid_scene_arrays(worlds)
    q0 = planner.default_joint_state.position.view(1, -1).repeat(B, 1)
    q0[:, 0] += torch.linspace(-0.3, 0.3, B, device=q0.device)  # a start state per problem
    cur = JointState.from_position(q0.clone(), planner.joint_names)
    goal_js = cur.clone()
    goal_js.position[:, 0] += torch.tensor([0.5, -0.5, 0.4, -0.4], device=q0.device)
    goal_js.position[:, 3] += 0.25
    goal = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    res = planner.plan_pose(goal, cur, max_attempts=2)
    assert res is not None
    ok = res.success[:, 0].cpu().numpy()
    assert ok.mean() >= 0.75, ok
    traj = res.js_solution.position[:, 0].cpu().numpy()
    env = np.arange(B)
    n = int(ok.sum())
    md = model.as_dict()
    H = traj.shape[1]
    np.testing.assert_allclose(traj[ok][:, 0], q0.cpu().numpy()[ok], atol=1e-4)
    chk = oracle.kinematics_forward(traj[ok].reshape(n * H, -1), md, horizon=H)
    s2 = chk["robot_spheres"].reshape(n, H, -1, 4)
    d = oracle.scene_collision(s2, arrays, 1.0, 0.0, env_query_idx=env[ok].astype(np.int32), use_multi_env=True)["distance"]
    assert (d.sum((1, 2)) == 0).all()
    np.testing.assert_allclose(chk["link_pos"].reshape(n, H, 3)[:, -1], goal.position[:, 0, 0, 0].cpu().numpy()[ok], atol=5e-3)
    res_c = planner.plan_cspace(goal_js, cur)
    okc = res_c.success[:, 0].cpu().numpy()
    assert okc.mean() >= 0.75
    np.testing.assert_allclose(res_c.js_solution.position[:, 0, -1].cpu().numpy(), goal_js.position.cpu().numpy(), atol=1e-4)
    # a smaller batch is padded with its first problem (reference :759-775) and sliced back
    small = planner.trajopt_solver.solve_cspace(JointState.from_position(goal_js.position[:2]), JointState.from_position(q0[:2]))
    assert small.success.shape == (2, 1) and small.js_solution.position.shape[0] == 2
