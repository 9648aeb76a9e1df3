"""The reference's public names (curobo.kinematics / collision_checking / rollout / optim / types /
inverse_kinematics) resolve in this repository's ``curobo`` namespace, and the call sequences of the reference's
own benchmark / tests run on them (GPU)."""

import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(autouse=True)
def _this_repos_curobo_namespace():
    """Other tests of the CPU suite import the REFERENCE's ``curobo`` package (live comparisons, /root/reference on
    sys.path): make ``import curobo`` resolve to this repository's namespace for the tests of this file."""
    stale = [m for m in sys.modules if (m == "curobo" or m.startswith("curobo.")) and not str(getattr(sys.modules[m], "__file__", "")).startswith(ROOT)]
    saved = {m: sys.modules.pop(m) for m in stale}
    path = list(sys.path)
    sys.path[:] = [ROOT] + [p for p in sys.path if p != ROOT]
    yield
    sys.path[:] = path
    for m in [m for m in sys.modules if m == "curobo" or m.startswith("curobo.")]:
        sys.modules.pop(m)
    sys.modules.update(saved)


def test_reference_public_names_resolve():
    from curobo import InverseKinematics, InverseKinematicsCfg  # noqa: F401  (reference: curobo/__init__.py)
    from curobo.collision_checking import RobotCollisionChecker, RobotCollisionCheckerCfg  # noqa: F401
    from curobo.kinematics import Kinematics, KinematicsCfg, KinematicsState  # noqa: F401
    from curobo.optim import MPPI, LBFGSOpt, LBFGSOptCfg, MPPICfg, MultiStageOptimizer  # noqa: F401
    from curobo.rollout import RosenbrockCfg, RosenbrockRollout  # noqa: F401
    from curobo.types import DeviceCfg, GoalToolPose, JointState, Pose, ToolPose, ToolPoseCriteria  # noqa: F401
    from curobo.scene import Capsule, Cuboid, Cylinder, Mesh, Obstacle, Scene, SceneData, Sphere, VoxelGrid  # noqa: F401  (reference: curobo/scene.py)
    from curobo.motion_planner import GraspPlanResult, MotionPlanner, MotionPlannerCfg  # noqa: F401
    from curobo.batch_motion_planner import BatchMotionPlanner  # noqa: F401
    from curobo.trajectory_optimizer import TrajectoryOptimizer, TrajectoryOptimizerCfg  # noqa: F401
    from curobo.model_predictive_control import (ModelPredictiveControl, ModelPredictiveControlCfg,  # noqa: F401
                                                 ModelPredictiveControlResult)
    import inspect

    # constructor shapes of the reference: LBFGSOpt(config, rollout_list, use_cuda_graph) (optim/gradient/lbfgs.py:156),
    # Kinematics(config, compute_jacobian, compute_spheres, compute_com) (robot/kinematics/kinematics.py:49),
    # compute_kinematics(joint_state, idxs_env) (:138)
    assert list(inspect.signature(LBFGSOpt.__init__).parameters)[1:4] == ["config", "rollout_list", "use_cuda_graph"]
    assert list(inspect.signature(Kinematics.__init__).parameters)[1:5] == ["config", "compute_jacobian", "compute_spheres", "compute_com"]
    assert list(inspect.signature(Kinematics.compute_kinematics).parameters)[1:3] == ["joint_state", "idxs_env"]
    for f in ("num_problems", "num_iters", "history", "step_scale", "use_cuda_kernel_step_direction", "use_cuda_kernel_shared_buffers",
              "use_cuda_kernel_line_search", "stable_mode", "solver_type", "line_search_scale"):
        assert f in LBFGSOptCfg.__dataclass_fields__, f  # the fields tests/_src/optim/gradient/test_lbfgs.py:249-261 sets


def test_rosenbrock_rollout_and_cost_containers():
    from curobo.rollout import Rollout, RosenbrockCfg, RosenbrockRollout
    from curobo.types import DeviceCfg

    ro = RosenbrockRollout(RosenbrockCfg(device_cfg=DeviceCfg("cpu"), dimensions=3))
    assert isinstance(ro, Rollout) and ro.action_dim == 3 and ro.action_horizon == 1
    x = torch.tensor([[[1.0, 1.0, 1.0]], [[0.0, 0.0, 0.0]], [[-1.0, 2.0, 0.5]]])
    c = ro.evaluate_action(x).costs_and_constraints.get_sum_cost_and_constraint(sum_horizon=True)
    want = [(1 - a) ** 2 + 100 * (b - a * a) ** 2 + (1 - b) ** 2 + 100 * (cc - b * b) ** 2 for a, b, cc in x[:, 0].tolist()]
    np.testing.assert_allclose(c.numpy(), want, rtol=1e-6)
    m = ro.compute_metrics_from_action(x)
    assert m.feasible.all() and m.convergence.shape == (3, 1)
    assert ro.get_initial_action(use_random=True).shape == (3, 1, 3)


def test_scene_config_in_the_reference_format():
    from curobo.scene import load_scene_config, scene_arrays_from_config

    cfg = {"cuboid": {"table": {"dims": [2.0, 2.0, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}},
           "sphere": {"ball": {"radius": 0.1, "pose": [0.4, 0, 0.4, 1, 0, 0, 0]}},
           "cylinder": {"post": {"radius": 0.05, "height": 0.8, "pose": [0.3, 0.3, 0.4, 1, 0, 0, 0]}},
           "capsule": {"bar": {"radius": 0.04, "base": [0, 0, 0], "tip": [0, 0, 0.4], "pose": [-0.3, 0.2, 0.3, 1, 0, 0, 0]}}}
    arr = scene_arrays_from_config(cfg)
    assert arr["cuboid_dims"].shape == (1, 4, 4) and arr["cuboid_dims"][0, :, 3].tolist() == [0.0, 1.0, 2.0, 3.0]
    np.testing.assert_allclose(arr["cuboid_dims"][0, 2], [0.04, 0.2, 0.0, 2.0])  # capsule: radius, half length
    np.testing.assert_allclose(arr["cuboid_inv_pose"][0, 2, :3], [0.3, -0.2, -0.5], atol=1e-6)  # centred on the segment
    assert load_scene_config("collision_table.yml")[0][0]["dims"] == [4.0, 4.0, 0.2]
    assert load_scene_config(None) is None


@pytest.mark.gpu
@pytest.mark.parametrize("use_cuda_graph", [False, True])
def test_reference_shaped_lbfgs_on_rollout_protocol_objects(use_cuda_graph, device):
    """the call sequence of the reference's tests/_src/optim/gradient/test_lbfgs.py:236-270 (create_optimizer,
    reinitialize, optimize) on a Rollout-protocol object written in torch (gradient through autograd), kernels = HIP"""
    from curobo.optim import LBFGSOpt, LBFGSOptCfg
    from curobo.rollout import CostsAndConstraints, RolloutResult, RosenbrockCfg, RosenbrockRollout
    from curobo.types import DeviceCfg

    class QuadraticRollout(RosenbrockRollout):  # the reference test's cost: sum (10 - x)^2 (MockRollout, :52-191)
        def __init__(self, dof, horizon):
            super().__init__(RosenbrockCfg(device_cfg=DeviceCfg(device), dimensions=dof, time_action_horizon=horizon, time_horizon=horizon))
            self._lows, self._highs = -torch.ones(dof, device=device), torch.ones(dof, device=device)

        def evaluate_action(self, act_seq, **kw):
            cc = CostsAndConstraints()
            cc.costs.add(((10.0 - act_seq) ** 2).sum(-1, keepdim=True), "cost")
            return RolloutResult(act_seq, None, cc)

    ro = QuadraticRollout(7, 4)
    cfg = LBFGSOptCfg(num_problems=4, num_iters=50, history=10, step_scale=0.98, use_cuda_kernel_step_direction=True,
                      use_cuda_kernel_shared_buffers=True, use_cuda_kernel_line_search=True, stable_mode=True, solver_type="lbfgs",
                      line_search_scale=[0, 0.1, 0.5, 1.0], inner_iters=25)
    opt = LBFGSOpt(cfg, [ro, ro], use_cuda_graph=use_cuda_graph)
    opt.update_num_problems(4)
    torch.manual_seed(0)
    x0 = torch.randn(4, 4, 7, device=device) * 10.0
    opt.reinitialize(x0.clone())
    out = opt.optimize(x0.clone()).clone()
    assert float(((10.0 - out) ** 2).sum(-1).mean()) < 1e-5  # the reference test's bar (:320)
    # the reference's optional convergence exit (fixed_iters = False): the same quadratic, 400 iterations allowed, stops once
    # more than 80 % of the problems carry the line-search kernel's convergence flag -- same answer, far fewer iterations
    cfg_e = LBFGSOptCfg(num_problems=4, num_iters=400, history=10, line_search_scale=[0, 0.1, 0.5, 1.0], inner_iters=25, fixed_iters=False)
    opt_e = LBFGSOpt(cfg_e, [ro], use_cuda_graph=use_cuda_graph)
    out_e = opt_e.optimize(x0.clone()).clone()
    assert float(((10.0 - out_e) ** 2).sum(-1).mean()) < 1e-5
    assert 25 <= opt_e._opt.iterations_run < 400, opt_e._opt.iterations_run
    # Rosenbrock (curobo.rollout) from the bounds' corner
    rr = RosenbrockRollout(RosenbrockCfg(device_cfg=DeviceCfg(device), dimensions=2))
    opt2 = LBFGSOpt(LBFGSOptCfg(num_problems=8, num_iters=400, history=7, inner_iters=25), [rr], use_cuda_graph=use_cuda_graph)
    x = rr.get_initial_action(use_random=True)
    rr.update_batch_size(8)
    x = torch.linspace(-1.2, 1.5, 16, device=device).view(8, 1, 2)
    best = opt2.optimize(x)
    assert float((best.view(8, 2) - 1.0).abs().max()) < 2e-2 and opt2.solve_time > 0.0


@pytest.mark.gpu
def test_ik_benchmark_call_sequence(device):
    """reference benchmark/ik_benchmark.py:55-165, line for line on the curobo namespace of this repository"""
    from curobo.inverse_kinematics import InverseKinematics as IKSolver
    from curobo.inverse_kinematics import InverseKinematicsCfg as IKSolverCfg
    from curobo.types import DeviceCfg, JointState

    batch_size, num_seeds = 100, 8
    device_cfg = DeviceCfg()
    config = IKSolverCfg.create(
        robot="franka.yml", optimizer_configs=["ik/lbfgs_ik.yml"], metrics_rollout="metrics_base.yml",
        transition_model="ik/transition_ik.yml", scene_model="collision_table.yml", use_cuda_graph=True, num_seeds=num_seeds,
        position_tolerance=0.005, optimizer_collision_activation_distance=0.0025, self_collision_check=True, device_cfg=device_cfg,
        override_iters_for_multi_link_ik=None, seed_solver_num_seeds=max(32, num_seeds * 2), max_batch_size=batch_size)
    ik_solver = IKSolver(config)
    q_sample = ik_solver.sample_configs(batch_size, rejection_ratio=10)
    assert q_sample.shape == (batch_size, 7)
    ik_solver.config.exit_early = False
    results = []
    for exit_early in (False, True):
        ik_solver.config.exit_early = exit_early
        ik_solver.reset_seed()
        kin_state = ik_solver.compute_kinematics(JointState.from_position(q_sample))
        goal_tool_poses = kin_state.tool_poses.as_goal()
        result = ik_solver.solve_pose(goal_tool_poses=goal_tool_poses, seed_config=None)
        success = 100.0 * torch.count_nonzero(result.success).item() / len(q_sample)
        p_err = np.percentile(result.position_error[result.success.view(-1)].cpu().numpy(), 90).item()
        q_err = np.percentile(result.rotation_error[result.success.view(-1)].cpu().numpy(), 90).item()
        results.append((success, p_err, q_err, result.solve_time))
        assert success >= 95.0 and p_err < 0.005 and q_err < 0.05, results
        # the solutions really solve the problem: FK of the solution reaches the goal
        st = ik_solver.compute_kinematics(result.js_solution[:, 0])
        err = (st.tool_poses.position[:, 0, 0] - goal_tool_poses.position[:, 0, 0, 0]).norm(dim=-1)
        assert float(err[result.success.view(-1)].max()) < 0.005


def test_pose_goal_tool_pose_and_criteria_types():
    """reference shapes and factories: Pose.from_list / multiply (_src/types/pose.py), GoalToolPose [batch, horizon, links,
    goalset, 3 | 4] with from_poses (_src/types/tool_pose.py:182-277), ToolPoseCriteria factories
    (_src/cost/tool_pose_criteria.py:143-200)"""
    from scipy.spatial.transform import Rotation as R

    from curobo.types import GoalToolPose, Pose, ToolPoseCriteria

    rng = np.random.default_rng(0)
    qa, qb = R.random(5, random_state=1), R.random(5, random_state=2)
    pa, pb = rng.normal(size=(5, 3)), rng.normal(size=(5, 3))
    wxyz = lambda r: torch.as_tensor(np.roll(r.as_quat(), 1, axis=-1), dtype=torch.float32)  # noqa: E731
    a = Pose(torch.as_tensor(pa, dtype=torch.float32), wxyz(qa))
    b = Pose(torch.as_tensor(pb, dtype=torch.float32), wxyz(qb))
    ab = a.multiply(b)
    np.testing.assert_allclose(ab.position.numpy(), pa + qa.apply(pb), atol=1e-5)
    want_q = np.roll((qa * qb).as_quat(), 1, axis=-1)
    got_q = ab.quaternion.numpy()
    assert np.minimum(np.abs(got_q - want_q).max(1), np.abs(got_q + want_q).max(1)).max() < 1e-5
    ident = a.multiply(a.inverse())
    np.testing.assert_allclose(ident.position.numpy(), 0.0, atol=1e-5)
    np.testing.assert_allclose(ident.quaternion.abs().numpy(), [[1, 0, 0, 0]] * 5, atol=1e-5)
    off = Pose.from_list([0, 0, -0.15, 1, 0, 0, 0])
    assert off.position.shape == (1, 3) and off.quaternion.shape == (1, 4)
    np.testing.assert_allclose(a.multiply(off).position.numpy(), pa + qa.apply([0, 0, -0.15]), atol=1e-5)  # in the tool frame
    np.testing.assert_allclose(off.multiply(a).position.numpy(), pa + [0, 0, -0.15], atol=1e-5)  # in the world frame

    g = GoalToolPose.from_poses({"tool": Pose(torch.arange(18.0).view(6, 3), torch.tensor([[1.0, 0, 0, 0]]).repeat(6, 1))}, num_goalset=3)
    assert g.position.shape == (2, 1, 1, 3, 3) and g.quaternion.shape == (2, 1, 1, 3, 4)
    assert (g.batch_size, g.horizon, g.num_links, g.num_goalset) == (2, 1, 1, 3) and len(g) == 1
    assert g.position[1, 0, 0, 2].tolist() == [15.0, 16.0, 17.0]
    p, q = g.static_goals()
    assert p.shape == (2, 1, 3, 3) and q.shape == (2, 1, 3, 4)
    assert g["tool"].position.shape == (6, 3) and g[1].position.shape == (1, 1, 1, 3, 3)
    two = GoalToolPose.from_poses({"a": Pose(torch.zeros(2, 3), torch.zeros(2, 4)), "b": Pose(torch.ones(2, 3), torch.ones(2, 4))},
                                  ordered_tool_frames=["b", "a"])
    assert two.tool_frames == ["b", "a"] and float(two.position[0, 0, 0, 0, 0]) == 1.0
    with pytest.raises(ValueError, match="5D"):
        GoalToolPose(["tool"], torch.zeros(2, 1, 1, 3), torch.zeros(2, 1, 1, 4))
    with pytest.raises(ValueError, match="num_links"):
        GoalToolPose(["a", "b"], torch.zeros(2, 1, 1, 1, 3), torch.zeros(2, 1, 1, 1, 4))
    with pytest.raises(ValueError, match="Missing poses"):
        GoalToolPose.from_poses({"a": off}, ordered_tool_frames=["a", "b"])

    c = ToolPoseCriteria()
    assert c.terminal_pose_axes_weight_factor == [1.0] * 6 and c.non_terminal_pose_axes_weight_factor == [0.0] * 6
    assert c.terminal_pose_convergence_tolerance == [0.0, 0.0] and c.project_distance_to_goal is False
    lm = ToolPoseCriteria.linear_motion("y", non_terminal_scale=2.0)
    assert lm.non_terminal_pose_axes_weight_factor == [2.0, 0.0, 2.0, 2.0, 2.0, 2.0] and lm.project_distance_to_goal is True
    assert lm.terminal_pose_axes_weight_factor == [1.0] * 6
    assert ToolPoseCriteria.track_position([1, 2, 3]).non_terminal_pose_axes_weight_factor == [1.0, 2.0, 3.0, 0.0, 0.0, 0.0]
    assert ToolPoseCriteria.track_orientation([1, 1, 1], 0.5).non_terminal_pose_axes_weight_factor == [0.0, 0.0, 0.0, 0.5, 0.5, 0.5]
    assert ToolPoseCriteria.track_position_and_orientation().non_terminal_pose_axes_weight_factor == [0.1] * 6
    with pytest.raises(ValueError, match="6 floats"):
        ToolPoseCriteria(terminal_pose_axes_weight_factor=[1.0, 1.0])
    with pytest.raises(ValueError, match="Invalid axis"):
        ToolPoseCriteria.linear_motion("w")
    assert lm.clone() == lm


def test_link_sphere_toggles_and_grasp_links_from_the_robot_file():
    """reference KinematicsParams.disable_link_spheres / enable_link_spheres / reset_link_spheres (robot/types/
    kinematics_params.py:558-595) and ``grasp_contact_link_names`` of the robot yaml (franka.yml:6-10)"""
    from curobo.kinematics import KinematicsCfg

    cfg = KinematicsCfg.from_packaged("franka", device="cpu")
    kp = cfg.kinematics_config
    assert kp.grasp_contact_link_names == ["panda_hand", "panda_leftfinger", "panda_rightfinger", "attached_object"]
    idx = kp.get_sphere_index_from_link_name("panda_leftfinger")
    assert idx.numel() > 0 and (kp.link_sphere_idx_map[idx].long() == kp.link_names.index("panda_leftfinger")).all()
    before = kp.link_spheres.clone()
    kp.disable_link_spheres("panda_leftfinger")
    assert (kp.link_spheres[:, idx, 3] == -100.0).all()
    other = torch.ones(kp.num_spheres, dtype=torch.bool)
    other[idx] = False
    assert torch.equal(kp.link_spheres[:, other], before[:, other]) and torch.equal(kp.link_spheres[:, idx, :3], before[:, idx, :3])
    kp.link_spheres[:, idx, :3] += 0.01
    kp.enable_link_spheres("panda_leftfinger")  # radius only
    assert torch.equal(kp.link_spheres[:, idx, 3], before[:, idx, 3]) and not torch.equal(kp.link_spheres[:, idx, :3], before[:, idx, :3])
    kp.reset_link_spheres("panda_leftfinger")
    assert torch.equal(kp.link_spheres, before)
    with pytest.raises(ValueError, match="not found"):
        kp.enable_link_spheres("nope")


def test_scene_types_build_the_same_stores_as_the_dictionary_format():
    """reference ``curobo.scene``: ``Scene`` (SceneCfg, geom/types.py:918-1292) of ``Cuboid`` / ``Sphere`` / ``Capsule`` / ``Cylinder`` /
    ``Mesh`` / ``VoxelGrid`` objects; ``create`` from the yaml dictionary, add / get / remove, box approximation"""
    from curobo.scene import Capsule, Cuboid, Cylinder, Mesh, Scene, Sphere, VoxelGrid, scene_arrays_from_config
    from curobo_amd.scene.config import mesh_envs_from_config, voxel_arrays_from_config

    cfg = {"cuboid": {"table": {"dims": [2.0, 2.0, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}},
           "sphere": {"ball": {"radius": 0.1, "pose": [0.4, 0, 0.4, 1, 0, 0, 0]}},
           "cylinder": {"post": {"radius": 0.05, "height": 0.8, "pose": [0.3, 0.3, 0.4, 1, 0, 0, 0]}},
           "capsule": {"bar": {"radius": 0.04, "base": [0, 0, 0], "tip": [0, 0, 0.4], "pose": [-0.3, 0.2, 0.3, 0.7071068, 0.7071068, 0, 0]}}}
    scene = Scene(cuboid=[Cuboid(name="table", dims=[2.0, 2.0, 0.2], pose=[0, 0, -0.1, 1, 0, 0, 0])],
                  sphere=[Sphere(name="ball", radius=0.1, pose=[0.4, 0, 0.4, 1, 0, 0, 0])])
    scene.add_obstacle(Cylinder(name="post", radius=0.05, height=0.8, pose=[0.3, 0.3, 0.4, 1, 0, 0, 0]))
    scene.add_obstacle(Capsule(name="bar", radius=0.04, base=[0, 0, 0], tip=[0, 0, 0.4], pose=[-0.3, 0.2, 0.3, 0.7071068, 0.7071068, 0, 0]))
    assert len(scene) == 4 and [o.name for o in scene] == ["ball", "table", "post", "bar"] and scene.get_cache_dict() == {"obb": 1, "mesh": 0}
    want, got, made = scene_arrays_from_config(cfg), scene_arrays_from_config(scene), scene_arrays_from_config(Scene.create(cfg))
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        np.testing.assert_array_equal(made[k], want[k], err_msg=k)
    assert scene.get_obstacle("post").height == 0.8 and scene.get_obstacle("nope") is None
    scene.remove_obstacle("ball")
    assert len(scene) == 3 and scene.sphere == [] and int(scene_arrays_from_config(scene)["cuboid_count"][0]) == 3
    two = scene_arrays_from_config([scene, Scene.create(cfg)])  # one world per environment
    assert two["cuboid_count"].tolist() == [3, 4]
    with pytest.raises(ValueError, match="requires a pose"):
        Cuboid(name="x", dims=[1, 1, 1])
    assert Sphere(name="s", radius=0.2, position=[1, 2, 3]).pose == [1, 2, 3, 1, 0, 0, 0]

    # boxes around the analytic kinds: every point of the obstacle is inside its box
    boxes = Scene.create(cfg).get_obb_world()
    assert len(boxes.cuboid) == 4 and len(boxes) == 4
    bar = [b for b in boxes.cuboid if b.name == "bar"][0]
    np.testing.assert_allclose(bar.dims, [0.08, 0.08, 0.48], atol=1e-6)
    np.testing.assert_allclose(bar.pose[:3], [-0.3, 0.2 - 0.2, 0.3], atol=1e-6)  # the axis (local z) lies along world -y... rotated 90 deg about x
    ball = [b for b in boxes.cuboid if b.name == "ball"][0]
    assert ball.dims == [0.2, 0.2, 0.2]

    # meshes and voxel grids
    v = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    f = [[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]]
    m = Mesh(name="tet", vertices=v, faces=f, scale=[0.5, 0.5, 2.0], pose=[0.1, 0, 0, 1, 0, 0, 0])
    assert m.scale is None and np.allclose(m.vertices[3], [0, 0, 2.0])
    np.testing.assert_allclose(m.get_cuboid().dims, [0.5, 0.5, 2.0])
    s2 = Scene(mesh=[m], cuboid=[Cuboid(name="floor", dims=[1, 1, 0.1], pose=[0, 0, -0.05, 1, 0, 0, 0])])
    envs = mesh_envs_from_config(s2)
    assert len(envs) == 1 and envs[0][0]["name"] == "tet" and np.asarray(envs[0][0]["vertices"]).shape == (4, 3)
    esdf = np.linspace(-1, 1, 4 * 3 * 2).astype(np.float32)
    grid = VoxelGrid(name="g", dims=[0.4, 0.3, 0.2], voxel_size=0.1, feature_tensor=torch.as_tensor(esdf).view(4, 3, 2), pose=[0, 0, 0.5, 1, 0, 0, 0])
    assert grid.get_grid_shape()[0] == [4, 3, 2]
    s2.add_obstacle(grid)
    va = voxel_arrays_from_config([s2, Scene()])
    assert va["voxel_count"].tolist() == [1, 0] and va["voxel_params"][0, 0].tolist() == [4.0, 3.0, 2.0, np.float32(0.1)]
    np.testing.assert_array_equal(va["voxel_features"][0, 0], esdf.astype(np.float16))
    np.testing.assert_allclose(va["voxel_inv_pose"][0, 0, :7], [0, 0, -0.5, 1, 0, 0, 0])
    with pytest.raises(ValueError, match="feature_tensor has"):
        voxel_arrays_from_config(Scene(voxel=[VoxelGrid(name="bad", dims=[0.4, 0.4, 0.4], voxel_size=0.1, feature_tensor=esdf, pose=[0, 0, 0, 1, 0, 0, 0])]))


def test_kinematics_accessors_and_result_helpers():
    """reference ``Kinematics`` members (robot/kinematics/kinematics.py): sizes, defaults, limits, active / full joint states with the
    robot file's locked joints; ``IKSolverResult.get_unique_solution``; the front ends' class surfaces"""
    from curobo import ModelPredictiveControl, ModelPredictiveControlCfg, TrajectoryOptimizer  # noqa: F401
    from curobo.inverse_kinematics import InverseKinematics
    from curobo.kinematics import Kinematics, KinematicsCfg
    from curobo.types import JointState
    from curobo_amd.solver.inverse_kinematics import InverseKinematicsResult

    k = Kinematics(KinematicsCfg.from_packaged("franka", device="cpu"))
    assert (k.dof, k.get_dof(), k.base_link, k.total_spheres) == (7, 7, "panda_link0", 65)
    np.testing.assert_allclose(k.default_joint_position.numpy(), [0.0, -1.3, 0.0, -2.5, 0.0, 1.5, 0.8], atol=1e-6)
    assert k.default_joint_state.joint_names == k.joint_names
    lim = k.get_joint_limits()
    assert lim.position.shape == (2, 7) and lim.velocity.shape == (2, 7) and lim.effort.shape == (7,) and lim.joint_names == k.joint_names
    lock = k.lock_jointstate
    assert lock.joint_names == ["panda_finger_joint1", "panda_finger_joint2"] and lock.position.tolist() == pytest.approx([0.04, 0.04])
    full = k.get_full_js(JointState.from_position(torch.arange(14.0).view(2, 7), joint_names=k.joint_names))
    assert full.position.shape == (2, 9) and full.joint_names[-2:] == lock.joint_names and full.position[1, -1].item() == pytest.approx(0.04)
    shuffled = JointState.from_position(full.position.flip(-1), joint_names=list(reversed(full.joint_names)))
    act = k.get_active_js(shuffled)
    assert act.joint_names == k.joint_names and torch.equal(act.position, torch.arange(14.0).view(2, 7))
    with pytest.raises(ValueError, match="lacks the active joints"):
        k.get_active_js(JointState.from_position(torch.zeros(1, 2), joint_names=["a", "b"]))
    assert k.get_self_collision_config().collision_pairs.shape[0] == 818
    # link offsets, the sphere set, in-place configuration updates (reference kinematics.py:345-366, 443-455, 476-478)
    assert k.robot_spheres.shape == (65, 4) and k.robot_spheres.data_ptr() == k.kinematics_config.link_spheres.data_ptr()
    kc = k.kinematics_config
    i3 = list(kc.link_names).index("panda_link3")
    t3 = k.get_link_transform("panda_link3")
    np.testing.assert_allclose(t3.position.numpy().reshape(3), kc.fixed_transforms[i3].reshape(3, 4)[:, 3].numpy(), atol=1e-7)
    all_t = k.get_all_link_transforms()
    assert all_t.position.shape == (kc.fixed_transforms.shape[0], 3) and all_t.quaternion.shape == (kc.fixed_transforms.shape[0], 4)
    R3 = kc.fixed_transforms[i3].reshape(3, 4)[:, :3].numpy()
    w, x, y, z = all_t.quaternion[i3].tolist()
    Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    np.testing.assert_allclose(Rq, R3, atol=1e-6)
    with pytest.raises(ValueError, match="not part of the kinematic model"):
        k.get_link_transform("no_such_link")
    import copy

    other = copy.deepcopy(kc)
    other.link_spheres[0, 3, 3] = 0.123
    ptr = kc.link_spheres.data_ptr()
    k.update_kinematics_config(other)
    assert kc.link_spheres.data_ptr() == ptr and kc.link_spheres[0, 3, 3].item() == pytest.approx(0.123)
    other.link_spheres = other.link_spheres[:, :-1]
    with pytest.raises(ValueError, match="changes shape"):
        k.update_kinematics_config(other)
    with pytest.raises(ValueError, match="must be > 0"):
        k.update_batch_size(0, 1)
    # articulated joints = active + locked; mimic joints follow their actuated joint (reference kinematics.py:313-315, 410-441)
    assert k.all_articulated_joint_names == k.joint_names + ["panda_finger_joint1", "panda_finger_joint2"]
    assert k.get_mimic_js(JointState.from_position(torch.zeros(1, 7), joint_names=k.joint_names)) is None
    import dataclasses as dc

    k.config.model = dc.replace(k.config.model, mimic_joints={k.joint_names[1]: [{"joint_name": "follower", "joint_offset": [-1.5, 0.5]}]})
    mj = k.get_mimic_js(JointState.from_position(torch.tensor([[0.0, 0.2, 0, 0, 0, 0, 0], [0.0, -1.0, 0, 0, 0, 0, 0]]), joint_names=k.joint_names))
    assert mj.joint_names == ["follower"] and mj.position.reshape(-1).tolist() == pytest.approx([0.2, 2.0])

    sol = torch.tensor([[[0.10, 0.2], [0.101, 0.2], [0.5, 0.5], [0.9, 0.9]]])
    res = InverseKinematicsResult(success=torch.tensor([[True, True, True, False]]), solution=sol, js_solution=None,
                                  position_error=torch.zeros(1, 4), rotation_error=torch.zeros(1, 4))
    uniq = res.get_unique_solution(roundoff_decimals=2)
    assert uniq.shape == (2, 2) and sorted(uniq[:, 0].tolist()) == pytest.approx([0.10, 0.5])
    for cls, names in ((InverseKinematics, ("solve_pose", "sample_configs", "update_world", "update_tool_pose_criteria", "update_link_inertial",
                                            "get_active_js", "get_full_js", "reset_seed", "reset_shape", "destroy", "default_joint_state")),
                       (TrajectoryOptimizer, ("solve_pose", "solve_cspace", "compute_trajectory_dt", "get_interpolated_trajectory",
                                              "update_tool_pose_criteria", "reset_seed", "destroy", "horizon", "opt_dim")),
                       (ModelPredictiveControl, ("setup", "update_goal_tool_poses", "update_current_state", "optimize_next_action",
                                                 "optimize_action_sequence", "reset_robot", "update_world", "update_tool_pose_criteria"))):
        for n in names:
            assert hasattr(cls, n), (cls.__name__, n)
    assert hasattr(ModelPredictiveControlCfg, "create")
    # enable / disable_tool_pose_tracking (reference solver_core.py:370-401) on every front end, through update_tool_pose_criteria
    from curobo import BatchMotionPlanner, MotionPlanner
    from curobo.types import ToolPoseCriteria
    from curobo_amd.solver.tracking import ToolPoseTrackingMixin

    for cls in (InverseKinematics, TrajectoryOptimizer, ModelPredictiveControl, MotionPlanner, BatchMotionPlanner):
        assert issubclass(cls, ToolPoseTrackingMixin)

    class Stub(ToolPoseTrackingMixin):
        tool_frames = ["left", "right"]

        def __init__(self):
            self._criteria = {}

        def update_tool_pose_criteria(self, c):
            self._criteria = dict(c)

    st = Stub()
    st.disable_tool_pose_tracking(["right"])
    assert list(st._criteria) == ["right"] and st._criteria["right"].terminal_pose_axes_weight_factor == [0.0] * 6
    st.enable_tool_pose_tracking(non_terminal_weight_factor=0.5)
    assert set(st._criteria) == {"left", "right"} and st._criteria["right"].terminal_pose_axes_weight_factor == [1.0] * 6
    assert st._criteria["left"].non_terminal_pose_axes_weight_factor == [0.5] * 6
    st.disable_tool_pose_tracking()
    assert all(c.terminal_pose_axes_weight_factor == ToolPoseCriteria.disabled().terminal_pose_axes_weight_factor for c in st._criteria.values())
    assert ModelPredictiveControl._tracking_non_terminal_factor == 1.0 and InverseKinematics._tracking_non_terminal_factor == 0.0


def test_solver_configurations_take_the_robot_files_acceleration_and_jerk_limits():
    """the trajectory optimiser and MPC front ends bound acceleration / jerk by the robot file's cspace values (reference:
    JointLimits.acceleration / .jerk from CSpaceParams): UR10e 12 / 500, Franka 15 / 500 (one value for all joints), and the
    G1's PER-JOINT lists as lists -- the rollouts' bound tensors are [2, dof]"""
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.model_predictive_control import ModelPredictiveControlCfg
    from curobo_amd.motion_planner import TrajectoryOptimizerCfg, robot_acceleration_jerk_limits
    from curobo_amd.rollout.trajopt_rollout import TrajOptRolloutCfg

    ur, fr, g1 = (KinematicsCfg.from_packaged(n, device="cpu") for n in ("ur10e", "franka", "unitree_g1"))
    assert robot_acceleration_jerk_limits(ur) == (12.0, 500.0)
    assert robot_acceleration_jerk_limits(fr) == (15.0, 500.0)
    assert robot_acceleration_jerk_limits(None) == (None, None)
    assert robot_acceleration_jerk_limits(g1) == (10.0, 500.0)  # (the G1 file lists 49 equal values)
    # a robot file with a different value per joint: the list goes through joint by joint
    import copy
    import dataclasses

    want_a = [8.0 + 0.25 * i for i in range(g1.model.num_dof)]
    model2 = dataclasses.replace(g1.model, cspace=dict(copy.deepcopy(g1.model.cspace), max_acceleration=want_a))
    g1 = dataclasses.replace(g1, model=model2)
    acc, jerk = robot_acceleration_jerk_limits(g1)
    assert acc == want_a and jerk == 500.0
    c = TrajectoryOptimizerCfg(kinematics=g1).solver_cfg()
    assert c.rollout.max_acceleration == want_a
    from curobo_amd.rollout.trajopt_rollout import joint_limit_vector

    v = joint_limit_vector(c.rollout.max_acceleration, g1.model.num_dof, "cpu")
    assert v.shape == (g1.model.num_dof,) and np.allclose(v.numpy(), want_a)
    assert joint_limit_vector(12.0, 6, "cpu").tolist() == [12.0] * 6
    with pytest.raises(ValueError, match="one per active joint"):
        joint_limit_vector([1.0, 2.0], 6, "cpu")
    with pytest.raises(ValueError, match="positive"):
        joint_limit_vector([1.0, 0.0, 1.0], 3, "cpu")
    c = TrajectoryOptimizerCfg(kinematics=ur).solver_cfg()
    assert (c.rollout.max_acceleration, c.rollout.max_jerk) == (12.0, 500.0)
    c = TrajectoryOptimizerCfg(kinematics=fr).solver_cfg()
    assert (c.rollout.max_acceleration, c.rollout.max_jerk) == (15.0, 500.0) == (TrajOptRolloutCfg().max_acceleration, TrajOptRolloutCfg().max_jerk)
    from curobo_amd.types import DeviceCfg

    m = ModelPredictiveControlCfg.create(ur, device_cfg=DeviceCfg(device="cpu"))
    assert (m.solver.rollout.max_acceleration, m.solver.rollout.max_jerk) == (12.0, 500.0)
