"""XRDF robot descriptions (the Isaac "extended robot description format": a yaml next to a URDF) -> the robot configuration
dictionary ``build_robot_model`` / ``KinematicsCfg.from_data_dict`` take.

Behaviour of the reference's ``convert_xrdf_to_curobo`` (curobo/_src/util/xrdf_util.py:24-176), which
``KinematicsCfg.from_robot_yaml_file`` calls for ``*.xrdf`` files: collision spheres and their buffers from the geometry block the
``collision`` section names (the ``self_collision`` section must name the same one), tool frames, the joints of ``cspace.joint_names``
active and every other actuated joint of the URDF locked at its default position (0 without one), per-joint acceleration / jerk limits
(locked joints get the largest active value), unit null-space and c-space distance weights, ``modifiers``: ``set_base_frame`` and
``add_frame`` (a fixed or jointed extra link); the base link defaults to the root link of the URDF."""

from __future__ import annotations

import copy
from typing import Dict, Optional, Union

from .urdf import load_urdf


def _required(d: Dict, key: str):
    if key not in d:
        raise ValueError(f"{key} key not found in xrdf")
    return d[key]


def convert_xrdf_to_config(xrdf: Union[str, Dict], urdf_path: str) -> Dict:
    """-> ``{"robot_cfg": {"kinematics": {...}}}`` (+ ``"dynamics"`` when the XRDF has such a block); ``urdf_path`` as it should
    appear in the result (the path ``build_robot_model`` / ``from_data_dict`` will read)"""
    if isinstance(xrdf, str):
        import yaml

        with open(xrdf) as fh:
            xrdf = yaml.safe_load(fh)
    if xrdf is None or _required(xrdf, "format") != "xrdf":
        raise ValueError("format is not xrdf")
    urdf = load_urdf(urdf_path)
    children = {j.child for j in urdf.joints.values()}
    base_link = next(name for name in urdf.links if name not in children)  # root of the tree
    actuated = [name for name, j in urdf.joints.items() if j.type != "fixed" and j.mimic_joint is None]
    out: Dict = {}
    if "collision" in xrdf:
        geometry = _required(xrdf["collision"], "geometry")
        block = _required(xrdf, "geometry")[geometry]
        if "spheres" not in block:
            raise ValueError("spheres key not found in xrdf")
        spheres = block["spheres"]
        out["collision_spheres"] = spheres
        buffer = xrdf["collision"].get("buffer_distance")
        out["collision_sphere_buffer"] = 0.0 if buffer is None else buffer
        out["collision_link_names"] = list(spheres.keys())
        if "self_collision" not in xrdf:
            raise ValueError("self_collision key not found in xrdf")
        if xrdf["self_collision"]["geometry"] != geometry:
            raise ValueError("self_collision geometry does not match collision geometry")
        out["self_collision_ignore"] = _required(xrdf["self_collision"], "ignore")
        sc_buffer = xrdf["self_collision"].get("buffer_distance")
        out["self_collision_buffer"] = {} if sc_buffer is None else sc_buffer
    out["tool_frames"] = copy.deepcopy(_required(xrdf, "tool_frames"))
    cspace = _required(xrdf, "cspace")
    active = list(_required(cspace, "joint_names"))
    defaults = _required(xrdf, "default_joint_positions")
    active_q = [defaults.get(j, 0.0) for j in actuated if j in active]
    locked = {j: defaults.get(j, 0.0) for j in actuated if j not in active}
    acc, jerk = list(_required(cspace, "acceleration_limits")), list(_required(cspace, "jerk_limits"))
    names = active + list(locked.keys())
    n_locked = len(names) - len(active)
    out["lock_joints"] = locked
    out["cspace"] = {
        "joint_names": names,
        "default_joint_position": active_q + list(locked.values()),
        "null_space_weight": [1.0] * len(names),
        "cspace_distance_weight": [1.0] * len(names),
        "max_acceleration": acc + [max(acc)] * n_locked,
        "max_jerk": jerk + [max(jerk)] * n_locked,
    }
    extra: Dict = {}
    for mod in xrdf.get("modifiers") or []:
        if len(mod) != 1:
            raise ValueError("Each modifier should have only one key")
        (kind, data), = mod.items()
        if kind == "set_base_frame":
            base_link = data
        elif kind == "add_frame":
            t = data["fixed_transform"]
            extra[data["frame_name"]] = {
                "parent_link_name": data["parent_frame_name"], "link_name": data["frame_name"], "joint_name": data["joint_name"],
                "joint_type": data["joint_type"],
                "fixed_transform": list(t["position"]) + [t["orientation"]["w"]] + list(t["orientation"]["xyz"]),
            }
        # (other modifiers: the reference warns and goes on)
    out["extra_links"] = extra
    out["base_link"] = base_link
    out["urdf_path"] = urdf_path
    result = {"robot_cfg": {"kinematics": out}}
    if "dynamics" in xrdf:
        result["robot_cfg"]["dynamics"] = xrdf["dynamics"]
    return result
