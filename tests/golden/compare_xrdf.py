"""``curobo_amd.robot.xrdf.convert_xrdf_to_config`` against the reference's ``convert_xrdf_to_curobo`` (util/xrdf_util.py) on the XRDF
file it ships (ur10e.xrdf), with and without extra modifiers / a joint left out of the cspace -- the dictionaries must be equal --
and the robot model built from the XRDF against the reference's loader on the converted dictionary.

    python tests/golden/compare_xrdf.py        (needs /root/reference)"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402
import yaml  # noqa: E402
from curobo._src.types.content_path import ContentPath  # noqa: E402
from curobo._src.util.xrdf_util import convert_xrdf_to_curobo  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.kinematics import KinematicsCfg  # noqa: E402
from curobo_amd.robot.xrdf import convert_xrdf_to_config  # noqa: E402

CONTENT = os.path.join(R.REF, "curobo", "content")
xrdf_path = os.path.join(CONTENT, "configs", "robot", "ur10e.xrdf")
urdf_path = os.path.join(CONTENT, "assets", "robot", "ur_description", "ur10e.urdf")
base = yaml.safe_load(open(xrdf_path))
variants = {"as shipped": base}
v = copy.deepcopy(base)
v["modifiers"].append({"add_frame": {"frame_name": "camera", "parent_frame_name": "tool0", "joint_name": "camera_joint", "joint_type": "fixed",
                                     "fixed_transform": {"position": [0.0, 0.05, 0.1], "orientation": {"w": 1.0, "xyz": [0.0, 0.0, 0.0]}}}})
variants["with an added frame"] = v
v = copy.deepcopy(base)
v["cspace"]["joint_names"] = v["cspace"]["joint_names"][:5]  # wrist_3 locked at its default
v["cspace"]["acceleration_limits"], v["cspace"]["jerk_limits"] = [12.0, 11.0, 10.0, 9.0, 8.0], [500.0, 400.0, 300.0, 200.0, 100.0]
variants["a joint left out of the cspace"] = v
ok = True
for label, x in variants.items():
    ref = convert_xrdf_to_curobo(ContentPath(robot_xrdf_absolute_path=xrdf_path, robot_urdf_absolute_path=urdf_path), input_xrdf_dict=copy.deepcopy(x))
    ours = convert_xrdf_to_config(copy.deepcopy(x), urdf_path)
    same = ref == ours
    ok &= same
    print(f"{label}: {'ok' if same else 'DIFFERENT'}")
    if not same:
        a, b = ref["robot_cfg"]["kinematics"], ours["robot_cfg"]["kinematics"]
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print("   ", k, "ref:", str(a.get(k))[:200], "| ours:", str(b.get(k))[:200])
m = KinematicsCfg.from_xrdf(xrdf_path, urdf_path, device="cpu").model
kc, sc = R.reference_kinematics_from_dict(convert_xrdf_to_config(xrdf_path, urdf_path)["robot_cfg"]["kinematics"])
npy = lambda t: t.detach().cpu().numpy()  # noqa: E731
same = list(kc.joint_names) == list(m.joint_names) and list(kc.link_name_to_idx_map.keys()) == list(m.link_names)
for f in ("fixed_transforms", "link_map", "joint_map", "joint_map_type", "joint_offset_map", "tool_frame_map", "link_sphere_idx_map", "link_spheres", "link_masses_com"):
    a_, b_ = npy(getattr(kc, f)), getattr(m, f)
    same &= a_.size == b_.size and np.allclose(a_.reshape(b_.shape).astype(np.float64), b_.astype(np.float64), rtol=0, atol=1e-6)
same &= np.array_equal(npy(sc.collision_pairs).astype(np.int64), m.collision_pairs.astype(np.int64)) and np.allclose(npy(sc.sphere_padding), m.sphere_padding)
same &= np.allclose(npy(kc.joint_limits.position), m.joint_limits_position, atol=1e-6)
print(f"model built from ur10e.xrdf against the reference's loader on the converted dictionary: {'ok' if same else 'DIFFERENT'}")
sys.exit(0 if ok and same else 1)
