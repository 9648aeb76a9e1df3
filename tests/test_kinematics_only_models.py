"""The packaged ``<robot>_kinematics_only`` models (tools/make_robot_fixtures.py --kinematics-only): the robot files as the
reference's IK benchmark loads them for its `IK` rows -- ``collision_link_names = None`` and ``lock_joints = None``
(benchmark/ik_benchmark.py:60-65).  Held on the CPU: no enabled sphere, no collision pair, the same tool frames, and the tool
poses of the full model at the same joint values (through the oracle's FK)."""
import numpy as np
import pytest

from conftest import load_model, sample_q


@pytest.mark.parametrize("robot", ["franka", "dual_ur10e", "unitree_g1"])
def test_kinematics_only_model_is_the_full_model_without_spheres_and_locks(robot, oracle):
    full, lean = load_model(robot), load_model(f"{robot}_kinematics_only")
    assert lean.tool_frames == full.tool_frames
    assert lean.collision_pairs.shape[0] == 0
    assert lean.num_spheres == 1 and (lean.link_spheres[..., 3] < 0).all()  # one disabled placeholder, nothing to collide
    assert lean.num_dof <= full.num_dof and lean.num_links <= full.num_links
    names_full, names_lean = list(full.joint_names), list(lean.joint_names)
    assert set(names_lean) <= set(names_full)
    q_lean = sample_q(lean, 16, seed=3)
    # the joints the lean model dropped do not move its tool frames: any value does
    q_full = sample_q(full, 16, seed=4)
    for j, n in enumerate(names_lean):
        q_full[:, names_full.index(n)] = q_lean[:, j]
    a = oracle.kinematics_forward(q_full, full.as_dict(), horizon=1)
    b = oracle.kinematics_forward(q_lean, lean.as_dict(), horizon=1)
    np.testing.assert_allclose(b["link_pos"], a["link_pos"], atol=2e-6)
    dq = np.abs(np.einsum("...i,...i->...", a["link_quat"], b["link_quat"]))
    np.testing.assert_allclose(dq, 1.0, atol=2e-6)
