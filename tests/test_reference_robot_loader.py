"""The robot loader (robot YAML + URDF -> kernel tensors) against the reference's own loader run on the CPU
(tests/golden/reference_robot_loader.py: the reference's ``UrdfRobotParser`` + ``KinematicsLoader``, unmodified, over small
stand-ins for yourdfpy / warp and its FK kernel through oracle/_ref) on every robot file the reference ships, and the packaged
fixtures against a fresh load.  Runs in a subprocess: the reference's ``curobo`` package shadows the repository's facade."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONTENT = "/root/reference/curobo/content"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_CONTENT), reason="needs the reference checkout")


@needs_reference
def test_loader_builds_the_tensors_the_reference_loader_builds():
    """franka (locked finger joints, attached-object placeholders), ur10e (position-limit clip), dual_ur10e (two tool frames),
    simple_mimic_robot (mimic joints, no spheres), both Unitree G1 files (49 / 35 dof, four tool frames, 162 k pairs)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libcurobo_ref.so")):
        pytest.skip("oracle/_ref is not built (python __graft_entry__.py)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_robot_loader.py")], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    lines = [l for l in out.stdout.splitlines() if ": ok" in l]
    assert len(lines) == 6 + 1, text[-3000:]  # (+ the Franka with 100 attached-object sphere slots of the reference's attachment tests)


@needs_reference
@pytest.mark.parametrize("name", ["franka", "ur10e", "unitree_g1"])
def test_packaged_fixtures_are_a_fresh_load(name):
    """the .npz files shipped under curobo_amd/content/robot are what the loader builds today from the reference's files
    (centre of mass as the reference composes it, limits after the cspace clip)"""
    from curobo_amd.robot import load_packaged_robot, load_robot_model

    a = load_packaged_robot(name)
    b = load_robot_model(os.path.join(REF_CONTENT, "configs", "robot", f"{name}.yml"), os.path.join(REF_CONTENT, "assets"))
    for k in a._ARRAY_FIELDS:
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.joint_names == b.joint_names and a.link_names == b.link_names and a.tool_frames == b.tool_frames


def test_inertial_frame_and_cspace_limits_follow_the_reference():
    """the two places where the reference's loader does more than read the URDF: the centre of mass is the position of
    (R, t) o (I, t) = R t + t (parser_urdf.py:157-170), and the cspace block clips the position range and scales the
    velocity range (kinematics_loader.py:1102-1124)"""
    from curobo_amd.robot import load_packaged_robot

    ur = load_packaged_robot("ur10e")
    # upper arm of the UR10e: inertial origin xyz (-0.306, 0, 0.175), rpy (0, pi/2, 0): R t = (0.175, 0, 0.306)
    i = ur.link_names.index("upper_arm_link")
    np.testing.assert_allclose(ur.link_masses_com[i], [-0.131, 0.0, 0.481, 12.93], atol=1e-6)
    # shoulder_pan: +-2 pi in the URDF, clipped by cspace.position_limit_clip = 0.1
    np.testing.assert_allclose(ur.joint_limits_position[:, 0], [-2 * np.pi + 0.1, 2 * np.pi - 0.1], atol=1e-6)


@needs_reference
def test_cuboid_store_of_a_scene_description_is_the_reference_cuboid_data():
    """scene yaml / dictionary -> dims, inverse poses, enable flags, counts: the reference's ``SceneCfg.create`` +
    ``CuboidData.from_scene_cfg`` / ``from_batch_scene_cfg`` run on the CPU, on its four scene files and on random rotated cuboids
    in two environments; and two rotated fp16 voxel grids against its ``VoxelData.from_scene_cfg``"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_scene_config.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert sum(": ok" in l for l in out.stdout.splitlines()) == 7, text[-3000:]


@needs_reference
def test_tool_pose_criteria_factories_are_the_references():
    """``ToolPoseCriteria.track_position / track_orientation / track_position_and_orientation / linear_motion / disabled``: axis
    factors, tolerances and the projection flag against the reference's class (cost/tool_pose_criteria.py) run on the CPU"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_tool_pose_criteria.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "factory calls: ok" in out.stdout, (out.stdout + out.stderr)[-2000:]


@needs_reference
def test_pose_multiply_inverse_and_from_matrix_are_the_references():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_pose_ops.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count(": ok") == 4 + 21 and "DIFFERENT" not in out.stdout, (out.stdout + out.stderr)[-2000:]


@needs_reference
def test_joint_state_members_are_the_references():
    """constructors, indexing, stack / cat / repeat, time scaling and finite differences, reorder / append / augment with
    lock joints, seed gathers, trajectory trims and the copy family of ``JointState`` against the reference's class"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_joint_state.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count(": ok") == 50 and "DIFFERENT" not in out.stdout, (out.stdout + out.stderr)[-2000:]


@needs_reference
def test_goal_tool_pose_from_poses_is_the_references():
    """frame order, goal-set layout [batch, 1, frames, goal set, 3 | 4] of the goals the pose cost reads"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_goal_tool_pose.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count(": ok") == 4 + 2, (out.stdout + out.stderr)[-2000:]


@needs_reference
def test_xrdf_conversion_is_the_references():
    """``*.xrdf`` robot descriptions: the configuration dictionary against the reference's ``convert_xrdf_to_curobo`` (its ur10e.xrdf
    as shipped, with an added frame, with a joint left out of the cspace) and the model built from it against its loader"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libcurobo_ref.so")):
        pytest.skip("oracle/_ref is not built (python __graft_entry__.py)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_xrdf.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count(": ok") == 4, (out.stdout + out.stderr)[-2000:]


@needs_reference
def test_deceleration_seeds_are_the_references():
    """``util/deceleration.py`` (knots that bring a moving robot to rest: the MPC's fallback seeds) against the reference's
    ``TrajectorySeedGenerator.generate_deceleration_seeds`` on random states: three profiles + an unknown name, two time steps, resting joints"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "compare_deceleration_seeds.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count(": ok") == 24 and "DIFFERENT" not in out.stdout, (out.stdout + out.stderr)[-2000:]
