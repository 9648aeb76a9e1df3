"""curobo_amd.robot.loader against THE REFERENCE'S OWN loader (tests/golden/reference_robot_loader.py) on the reference's robot
files: every kernel tensor of ``KinematicsParams``, the joint limits and the self-collision pair list.

    python tests/golden/compare_robot_loader.py [robot ...]        (needs /root/reference; prints one line per robot)

Integer tables and names must be identical; float tensors agree to 1e-6 (the links behind locked joints get their fixed
transform from an FK evaluation in fp32 on the reference's side).  A robot without collision spheres: the reference keeps a
[1, 1, 4] placeholder of zeros with ``total_spheres = 0``, this loader an empty [1, 0, 4] array."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402  (puts /root/reference ahead of the repository on the path)

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.robot import load_robot_model  # noqa: E402

CONTENT = os.path.join(R.REF, "curobo", "content")
ROBOTS = ["franka", "ur10e", "dual_ur10e", "simple_mimic_robot", "unitree_g1_29dof_retarget", "unitree_g1"]


def npy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def with_attached_object_slots(name, n):
    """the robot file with ``extra_collision_spheres: {attached_object: n}`` (what the reference's attachment tests load,
    tests/_src/collision/test_attachment_manager.py:25-31), written next to nothing of the reference's: a temporary file"""
    import tempfile

    import yaml

    with open(os.path.join(CONTENT, "configs", "robot", f"{name}.yml")) as fh:
        d = yaml.safe_load(fh)
    d["robot_cfg"]["kinematics"]["extra_collision_spheres"] = {"attached_object": n}
    fd, path = tempfile.mkstemp(suffix=f"_{name}_attached_{n}.yml")
    with os.fdopen(fd, "w") as fh:
        yaml.safe_dump(d, fh)
    return path


def compare(name):
    yml = name if name.endswith(".yml") else os.path.join(CONTENT, "configs", "robot", f"{name}.yml")
    kc, sc = R.reference_kinematics(yml)
    m = load_robot_model(yml, os.path.join(CONTENT, "assets"))
    bad = []
    for what, a, b in (("link names", list(kc.link_name_to_idx_map.keys()), list(m.link_names)), ("joint names", list(kc.joint_names), list(m.joint_names)),
                       ("tool frames", list(kc.tool_frames), list(m.tool_frames)), ("num_dof", int(kc.num_dof), int(m.num_dof))):
        if a != b:
            bad.append(f"{what}: {a} != {b}")
    no_spheres = int(kc.total_spheres) == 0
    for f in ("fixed_transforms", "link_map", "joint_map", "joint_map_type", "joint_offset_map", "tool_frame_map", "link_sphere_idx_map",
              "link_chain_data", "link_chain_offsets", "joint_links_data", "joint_links_offsets", "joint_affects_endeffector", "link_spheres",
              "link_masses_com", "link_inertias"):
        a, b = npy(getattr(kc, f)), getattr(m, f)
        if no_spheres and f in ("link_sphere_idx_map", "link_spheres"):
            if b.size != 0 or np.abs(a).max() != 0:
                bad.append(f"{f}: expected the reference's zero placeholder and an empty array")
            continue
        if a.size != b.size:
            bad.append(f"{f}: shape {a.shape} != {b.shape}")
            continue
        a = a.reshape(b.shape)
        if a.dtype.kind == "f":
            if not np.allclose(a, b, rtol=0, atol=1e-6):
                bad.append(f"{f}: max |diff| {np.abs(a.astype(np.float64) - b).max():.3e}")
        elif not np.array_equal(a.astype(np.int64), b.astype(np.int64)):
            bad.append(f"{f}: differs")
    # the tree levels the RNEA kernels walk (kinematics_params.py:250-290), derived from the link map on both sides
    import torch
    from curobo_amd.robot.kinematics_params import KinematicsParams

    kp = KinematicsParams.from_model(m, torch.device("cpu"))
    for f in ("link_level_data", "link_level_offsets"):
        if not np.array_equal(npy(getattr(kc, f)).astype(np.int64), npy(getattr(kp, f)).astype(np.int64)):
            bad.append(f"{f}: differs")
    if int(kc.max_level_width) != int(kp.max_level_width) or int(kc.n_tree_levels) != int(kp.n_tree_levels):
        bad.append("tree level width / count differs")
    jl = kc.joint_limits
    for f, a, b in (("position limits", npy(jl.position), m.joint_limits_position), ("velocity limits", npy(jl.velocity), m.joint_limits_velocity)):
        if not np.allclose(a, b, rtol=0, atol=1e-6):
            bad.append(f"{f}: max |diff| {np.abs(a - b).max():.3e}")
    if getattr(jl, "effort", None) is not None and not np.allclose(npy(jl.effort)[1], m.joint_limits_effort, rtol=0, atol=1e-6):
        bad.append(f"effort limits: max |diff| {np.abs(npy(jl.effort)[1] - m.joint_limits_effort).max():.3e}")
    for f, key in (("acceleration", "max_acceleration"), ("jerk", "max_jerk")):
        if not np.allclose(npy(getattr(jl, f))[1], np.asarray(m.cspace[key], np.float32), rtol=0, atol=1e-6):
            bad.append(f"{f} limits (cspace {key}, per active joint) differ")
    dq = kc.cspace.default_joint_position
    if dq is not None and m.cspace.get("default_joint_position") is not None:
        if not np.allclose(npy(dq), np.asarray(m.cspace["default_joint_position"], np.float32), rtol=0, atol=1e-6):
            bad.append("cspace default joint position (reindexed to the active joints) differs")
    # mimic joints {actuated joint: [{joint_name, joint_offset}]} and the articulated joint names (active + locked)
    ref_mimic = {k: [(d["joint_name"], [float(x) for x in d["joint_offset"]]) for d in v] for k, v in (kc.mimic_joints or {}).items()}
    our_mimic = {k: [(d["joint_name"], [float(x) for x in d["joint_offset"]]) for d in v] for k, v in (m.mimic_joints or {}).items()}
    if set(ref_mimic) != set(our_mimic) or any(sorted(ref_mimic[k]) != sorted(our_mimic[k]) for k in ref_mimic):
        bad.append(f"mimic joints differ: {ref_mimic} vs {our_mimic}")
    if list(kc.non_fixed_joint_names) != list(m.joint_names) + list(m.lock_joints.keys()):
        bad.append(f"articulated joint names differ: {kc.non_fixed_joint_names} vs {list(m.joint_names) + list(m.lock_joints.keys())}")
    if sc is not None and not no_spheres:
        if not np.array_equal(npy(sc.collision_pairs).astype(np.int64), m.collision_pairs.astype(np.int64)):
            bad.append("self-collision pair list differs")
        if not np.allclose(npy(sc.sphere_padding), m.sphere_padding, rtol=0, atol=1e-7):
            bad.append("self-collision sphere padding differs")
    print(f"{name}: {'ok' if not bad else 'DIFFERENT: ' + '; '.join(bad)}  ({m.num_dof} dof, {m.num_links} links, {m.num_spheres} spheres, "
          f"{m.collision_pairs.shape[0]} pairs, {len(m.tool_frames)} tool frames)", flush=True)
    return not bad


if __name__ == "__main__":
    names = sys.argv[1:] or ROBOTS
    results = [compare(n) for n in names]
    if not sys.argv[1:]:  # the robot the reference's attachment tests load: Franka with 100 sphere slots on the attached-object link
        tmp = with_attached_object_slots("franka", 100)
        try:
            results.append(compare(tmp))
        finally:
            os.remove(tmp)
    sys.exit(0 if all(results) else 1)
