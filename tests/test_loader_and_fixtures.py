"""Host logic: URDF/YAML loader, packaged robot fixtures, scene stores, workloads."""

import os

import numpy as np
import pytest

from conftest import load_model

REF_CONTENT = "/root/reference/curobo/content"


def test_packaged_fixture_shapes(franka, ur10e, g1):
    assert (franka.num_dof, franka.num_links, franka.num_spheres, franka.collision_pairs.shape[0]) == (7, 13, 65, 818)
    assert (ur10e.num_dof, ur10e.num_spheres, ur10e.collision_pairs.shape[0]) == (6, 20, 83)
    assert (g1.num_dof, g1.num_links, g1.num_spheres, len(g1.tool_frames)) == (49, 56, 674, 4)
    assert g1.collision_pairs.shape[0] > 512 * 256  # the reference's map-reduce regime
    for m in (franka, ur10e, g1):
        assert (m.link_map[1:] < np.arange(1, m.num_links)).all(), "parents must precede children"
        assert m.link_chain_offsets[-1] == len(m.link_chain_data)
        assert m.joint_map.max() == m.num_dof - 1
        assert m.collision_pairs.dtype == np.int16 and (m.collision_pairs[:, 0] < m.collision_pairs[:, 1]).all()
        assert m.fixed_transforms.dtype == np.float32


def test_franka_locked_fingers_are_fixed_links(franka):
    lf, rf = franka.link_names.index("panda_leftfinger"), franka.link_names.index("panda_rightfinger")
    assert franka.joint_map_type[lf] == -1 and franka.joint_map_type[rf] == -1
    assert franka.lock_joints == {"panda_finger_joint1": 0.04, "panda_finger_joint2": 0.04}
    # prismatic along +y / -y by 0.04 m from the hand frame
    assert franka.fixed_transforms[lf, 1, 3] == pytest.approx(0.04, abs=1e-6)
    assert franka.fixed_transforms[rf, 1, 3] == pytest.approx(-0.04, abs=1e-6)


@pytest.mark.skipif(not os.path.isdir(REF_CONTENT), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["franka", "ur10e"])
def test_loader_reproduces_packaged_fixture(name):
    from curobo_amd.robot import load_robot_model

    fresh = load_robot_model(f"{REF_CONTENT}/configs/robot/{name}.yml", f"{REF_CONTENT}/assets")
    packed = load_model(name)
    for k, v in fresh.as_dict().items():
        np.testing.assert_array_equal(v, packed.as_dict()[k], err_msg=k)
    assert fresh.joint_names == packed.joint_names and fresh.link_names == packed.link_names


def test_inverse_pose_roundtrip():
    from curobo_amd.scene import inverse_pose7

    p = [0.3, -0.2, 0.5, 0.8, 0.1, -0.5, 0.3]
    pi = inverse_pose7(p)
    back = inverse_pose7(pi)
    q = np.asarray(p[3:]) / np.linalg.norm(p[3:])
    np.testing.assert_allclose(back[:3], p[:3], atol=1e-12)
    np.testing.assert_allclose(back[3:], q, atol=1e-12)


def test_seed_knots_are_sharding_invariant(franka):
    from curobo_amd.workloads import seed_knots

    full = seed_knots(franka, 8, 12, seed=2)
    lo, hi = seed_knots(franka, 4, 12, seed=2, seed_offset=0), seed_knots(franka, 4, 12, seed=2, seed_offset=4)
    np.testing.assert_array_equal(full, np.concatenate([lo, hi]))
    assert (full >= franka.joint_limits_position[0]).all() and (full <= franka.joint_limits_position[1]).all()


def test_rollout_oracle_composition_runs(oracle, franka):
    """the oracle pipeline used by smoke()/bench cpu_baseline: finite, positive costs, FD-consistent"""
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration
    from oracle.rollout_ref import rollout_cost_and_gradient

    arrays = cuboid_scene_arrays(c2_world())
    knots = seed_knots(franka, 4, 12, seed=1)
    start = start_configuration(franka)
    kw = dict(use_sweep=False, use_speed_metric=False)  # smooth variant for the FD check
    r = rollout_cost_and_gradient(oracle, franka.as_dict(), arrays, knots, start, **kw)
    assert np.isfinite(r["cost"]).all() and (r["cost"] > 0).any()
    b = int(np.argmax(r["cost"]))
    g = r["grad_knots"][b]
    k, d = np.unravel_index(np.argmax(np.abs(g)), g.shape)
    eps = 2e-4
    kp, km = knots.copy(), knots.copy()
    kp[b, k, d] += eps
    km[b, k, d] -= eps
    fd = (rollout_cost_and_gradient(oracle, franka.as_dict(), arrays, kp, start, **kw)["cost"][b].astype(np.float64)
          - rollout_cost_and_gradient(oracle, franka.as_dict(), arrays, km, start, **kw)["cost"][b]) / (2 * eps)
    assert fd == pytest.approx(g[k, d], rel=0.15), (fd, g[k, d])


@pytest.mark.skipif(not os.path.isdir(REF_CONTENT), reason="reference checkout not present")
def test_kinematics_cfg_constructors(oracle, franka):
    """KinematicsCfg.from_data_dict / from_basic_urdf (reference kinematics_cfg.py:68-88,186-211):
    the dictionary route reproduces the packaged tensors; the URDF-only route gives the same tool pose."""
    import numpy as np
    import yaml

    from curobo_amd.kinematics import KinematicsCfg

    with open(f"{REF_CONTENT}/configs/robot/franka.yml") as fh:
        data = yaml.safe_load(fh)
    cfg = KinematicsCfg.from_data_dict(data, assets_root=f"{REF_CONTENT}/assets", device="cpu")
    for k in ("fixed_transforms", "link_map", "joint_map", "link_sphere_idx_map", "collision_pairs"):
        np.testing.assert_array_equal(np.asarray(getattr(cfg.model, k)), np.asarray(getattr(franka, k)), err_msg=k)
    kin = data["robot_cfg"]["kinematics"]
    urdf = os.path.join(f"{REF_CONTENT}/assets", kin["urdf_path"])
    basic = KinematicsCfg.from_basic_urdf(urdf, kin["base_link"], kin["tool_frames"], device="cpu")
    assert basic.model.num_spheres == 0
    # the URDF-only model keeps the finger joints free (no lock_joints): compare the arm chain at the locked values
    q7 = np.array([[0.1, -0.6, 0.3, -2.0, 0.2, 1.5, 0.7]], np.float32)
    pose_full = oracle.kinematics_forward(q7, franka.as_dict(), compute_spheres=False)
    names = list(basic.model.joint_names)
    qb = np.zeros((1, basic.model.num_dof), np.float32)
    for i, n in enumerate(franka.joint_names):
        qb[0, names.index(n)] = q7[0, i]
    for n, v in (kin.get("lock_joints") or {}).items():
        if n in names:
            qb[0, names.index(n)] = v
    pose_basic = oracle.kinematics_forward(qb, basic.model.as_dict(), compute_spheres=False)
    np.testing.assert_allclose(pose_basic["link_pos"], pose_full["link_pos"], atol=1e-6)
    np.testing.assert_allclose(np.abs(pose_basic["link_quat"]), np.abs(pose_full["link_quat"]), atol=1e-6)


def test_trajopt_seed_generation_host_logic():
    """TrajOptSolver.seed_goal_choice / seed_knots (pure torch; reference trajectory_seed_generator.py:
    122-170 linear interpolation start -> per-seed IK goal): checked without building any rollout."""
    import types

    import torch

    from curobo_amd.solver.trajopt import TrajOptSolver, TrajOptSolverCfg

    P, S, K, D, nk = 3, 6, 4, 5, 8
    slv = TrajOptSolver.__new__(TrajOptSolver)
    slv.P, slv.S, slv.K, slv.device = P, S, K, torch.device("cpu")
    slv.cfg = TrajOptSolverCfg(num_seeds=S, num_ik_goals=K)
    slv.cfg.rollout.n_knots = nk
    lim = torch.stack([-3.0 * torch.ones(D), 3.0 * torch.ones(D)])
    slv.kin = types.SimpleNamespace(num_dof=D, joint_limits_position=lim)
    ok = torch.tensor([[1, 1, 1, 1], [1, 0, 1, 0], [0, 0, 0, 0]], dtype=torch.bool)
    choice = slv.seed_goal_choice(ok)
    assert choice.tolist() == [[0, 1, 2, 3, 0, 1], [0, 0, 2, 0, 0, 0], [0, 0, 0, 0, 0, 0]]
    g = torch.Generator().manual_seed(0)
    goals = torch.rand(P, K, D, generator=g) * 2 - 1
    start = torch.rand(1, D, generator=g) - 0.5
    knots = slv.seed_knots(start, goals, choice)
    assert knots.shape == (P, S, nk, D)
    # (the free knots sit where the configured placement puts them: the reference's -- both ends included -- by default)
    assert slv.cfg.seed_knot_placement == "reference"
    t = torch.linspace(0, 1, nk).view(1, 1, nk, 1)
    sel = torch.gather(goals, 1, choice.unsqueeze(-1).expand(P, S, D))
    line = start.view(1, 1, 1, D) * (1 - t) + sel.view(P, S, 1, D) * t
    first = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 0, 1, 0, 0, 0], [1, 0, 0, 0, 0, 0]], dtype=torch.bool)
    dev = (knots - line).abs().amax((-1, -2))
    assert (dev[first] < 1e-6).all(), "the first seed aimed at a goal is the straight line"
    assert (dev[~first] > 1e-3).all(), "seeds that repeat a goal are perturbed"
    # every seed of a problem is distinct
    flat = knots.view(P, S, -1)
    for p in range(P):
        assert torch.cdist(flat[p], flat[p]).fill_diagonal_(1.0).min() > 1e-3
    # one shared goal (K = 1): seed 0 straight, the others bumped -- the original behaviour
    slv.K = 1
    c1 = slv.seed_goal_choice(ok[:, :1])
    assert int(c1.abs().sum()) == 0
    k1 = slv.seed_knots(start, goals[:, 0], c1)
    torch.testing.assert_close(k1[:, 0], (start.view(1, 1, D) * (1 - t[0]) + goals[:, 0].view(P, 1, D) * t[0]))


@pytest.mark.parametrize("robot", ["franka", "ur10e", "unitree_g1"])
def test_kinematics_params_from_model_on_cpu(robot):
    """RobotModel -> KinematicsParams (host tensors on the CPU device): every table the kernels read is
    there with the documented dtype / shape, including the effort limits of the torque-limit cost and the
    tree levels of RNEA"""
    import torch

    from conftest import load_model
    from curobo_amd.robot.kinematics_params import KinematicsParams

    model = load_model(robot)
    kin = KinematicsParams.from_model(model, torch.device("cpu"))
    D, L = kin.num_dof, kin.num_links
    assert kin.fixed_transforms.shape == (L, 3, 4) and kin.fixed_transforms.dtype == torch.float32
    assert kin.joint_limits_position.shape == (2, D) and kin.joint_limits_velocity.shape == (2, D)
    assert kin.joint_limits_effort is not None and kin.joint_limits_effort.shape == (D,) and bool((kin.joint_limits_effort > 0).all())
    assert kin.link_map.dtype == torch.int16 and kin.joint_map_type.dtype == torch.int8
    # parents precede children (the FK chain composes in index order), levels partition the links
    lm = kin.link_map.long()
    assert bool((lm[1:] < torch.arange(1, L)).all())
    lv = kin.link_level_offsets.long()
    assert int(lv[0]) == 0 and int(lv[-1]) == L and kin.n_tree_levels == lv.numel() - 1
    assert sorted(kin.link_level_data.long().tolist()) == list(range(L))
    pairs = kin.self_collision.collision_pairs
    assert pairs.shape[1] == 2 and int(pairs.max()) < kin.num_spheres


def test_kinematics_params_validate_shapes():
    """reference KinematicsParams.validate_shapes (robot/types/kinematics_params.py:212-248)"""
    import dataclasses

    import torch

    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams

    k = KinematicsParams.from_model(load_packaged_robot("franka"), torch.device("cpu"))
    k.validate_shapes()
    with pytest.raises(ValueError, match="link_chain_offsets"):
        dataclasses.replace(k, link_chain_offsets=k.link_chain_offsets[:-1]).validate_shapes()
    with pytest.raises(ValueError, match="joint_links_offsets"):
        dataclasses.replace(k, joint_links_offsets=k.joint_links_offsets[:-1]).validate_shapes()
    with pytest.raises(ValueError, match="joint_affects_endeffector"):
        dataclasses.replace(k, joint_affects_endeffector=k.joint_affects_endeffector.reshape(-1)[:-1]).validate_shapes()


def test_cspace_lists_are_validated_with_descriptive_errors():
    """cspace lists (velocity_scale, position_limit_clip, max_acceleration, max_jerk): one value, one per active joint, or one per
    cspace.joint_names entry (reindexed by name); anything else is a ValueError that names the key (a bare
    KeyError / a broadcast error)."""
    import os

    import yaml

    from curobo_amd.robot.loader import build_robot_model
    from curobo_amd.robot.urdf import load_urdf

    ref = "/root/reference/curobo/content"
    if not os.path.isdir(ref):
        pytest.skip("needs the reference's robot files (this container)")
    cfg = yaml.safe_load(open(os.path.join(ref, "configs", "robot", "ur10e.yml")))
    cfg = cfg.get("robot_cfg", cfg)["kinematics"]
    urdf = load_urdf(os.path.join(ref, "assets", cfg["urdf_path"]))
    base = build_robot_model(dict(cfg), urdf)
    D = base.num_dof

    def with_cspace(**kw):
        c = dict(cfg)
        c["cspace"] = dict(cfg["cspace"], **kw)
        return c

    m = build_robot_model(with_cspace(max_acceleration=[float(i + 1) for i in range(D)], joint_names=None), urdf)
    assert m.cspace["max_acceleration"] == [float(i + 1) for i in range(D)], "a list of length dof is taken as written"
    names = list(cfg["cspace"]["joint_names"])
    m = build_robot_model(with_cspace(max_jerk=[100.0 * (i + 1) for i in range(len(names))]), urdf)
    assert m.cspace["max_jerk"] == [100.0 * (names.index(n) + 1) for n in base.joint_names], "reindexed by name"
    with pytest.raises(ValueError, match="cspace.max_acceleration holds 3 values"):
        build_robot_model(with_cspace(max_acceleration=[1.0, 2.0, 3.0]), urdf)
    with pytest.raises(ValueError, match="cspace.position_limit_clip holds 2 values"):
        build_robot_model(with_cspace(position_limit_clip=[0.1, 0.2]), urdf)
    with pytest.raises(ValueError, match="does not list the active joint"):
        build_robot_model(with_cspace(velocity_scale=[0.5] * len(names), joint_names=["nope"] * len(names)), urdf)
