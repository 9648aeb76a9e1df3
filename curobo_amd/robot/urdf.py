"""Minimal URDF reader (xml.etree) for the kinematic quantities the hot path needs.

The reference delegates URDF parsing to ``yourdfpy`` (third party, absent here; reference
``curobo/_src/robot/parser/parser_urdf.py:14``).  Only joint origins / axes / limits / mimic tags and
link inertials are consumed by the kinematics loader (``parser_urdf.py:133-323``), so this module
reads exactly those.  Result parity is pinned by the reference's FK known-answer test.
"""

from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


def _floats(text: Optional[str], n: int, default: float = 0.0) -> List[float]:
    if text is None:
        return [default] * n
    vals = [float(x) for x in text.split()]
    assert len(vals) == n, f"expected {n} numbers, got {text!r}"
    return vals


def origin_to_matrix(xyz: List[float], rpy: List[float]) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw: R = Rz(yaw) @ Ry(pitch) @ Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    T = np.eye(4)
    T[:3, :3] = np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )
    T[:3, 3] = xyz
    return T


@dataclass
class UrdfJoint:
    name: str
    type: str
    parent: str
    child: str
    origin: np.ndarray
    axis: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.0])
    lower: Optional[float] = None
    upper: Optional[float] = None
    velocity: Optional[float] = None
    effort: Optional[float] = None
    mimic_joint: Optional[str] = None
    mimic_multiplier: float = 1.0
    mimic_offset: float = 0.0


@dataclass
class UrdfLink:
    name: str
    mass: Optional[float] = None
    inertial_origin: Optional[np.ndarray] = None
    inertia: Optional[np.ndarray] = None  # 3x3


@dataclass
class UrdfModel:
    links: Dict[str, UrdfLink]
    joints: Dict[str, UrdfJoint]  # insertion-ordered (file order)


def load_urdf(path: str) -> UrdfModel:
    root = ET.parse(path).getroot()
    links: Dict[str, UrdfLink] = {}
    for le in root.findall("link"):
        link = UrdfLink(name=le.attrib["name"])
        ie = le.find("inertial")
        if ie is not None:
            me = ie.find("mass")
            link.mass = float(me.attrib["value"]) if me is not None else 0.0
            oe = ie.find("origin")
            if oe is not None:
                link.inertial_origin = origin_to_matrix(
                    _floats(oe.attrib.get("xyz"), 3), _floats(oe.attrib.get("rpy"), 3)
                )
            ine = ie.find("inertia")
            if ine is not None:
                g = lambda k: float(ine.attrib.get(k, 0.0))  # noqa: E731
                link.inertia = np.array(
                    [
                        [g("ixx"), g("ixy"), g("ixz")],
                        [g("ixy"), g("iyy"), g("iyz")],
                        [g("ixz"), g("iyz"), g("izz")],
                    ]
                )
            else:
                link.inertia = np.zeros((3, 3))
        links[link.name] = link
    joints: Dict[str, UrdfJoint] = {}
    for je in root.findall("joint"):
        oe = je.find("origin")
        origin = (
            origin_to_matrix(_floats(oe.attrib.get("xyz"), 3), _floats(oe.attrib.get("rpy"), 3))
            if oe is not None
            else np.eye(4)
        )
        joint = UrdfJoint(
            name=je.attrib["name"],
            type=je.attrib["type"],
            parent=je.find("parent").attrib["link"],
            child=je.find("child").attrib["link"],
            origin=origin,
        )
        ae = je.find("axis")
        if ae is not None:
            joint.axis = _floats(ae.attrib.get("xyz"), 3)
        lim = je.find("limit")
        if lim is not None:
            for k in ("lower", "upper", "velocity", "effort"):
                if k in lim.attrib:
                    setattr(joint, k, float(lim.attrib[k]))
            # URDF defaults for revolute/prismatic limits when the attribute is absent
            if joint.lower is None:
                joint.lower = 0.0
            if joint.upper is None:
                joint.upper = 0.0
        mim = je.find("mimic")
        if mim is not None:
            joint.mimic_joint = mim.attrib["joint"]
            joint.mimic_multiplier = float(mim.attrib.get("multiplier", 1.0))
            joint.mimic_offset = float(mim.attrib.get("offset", 0.0))
        joints[joint.name] = joint
    return UrdfModel(links=links, joints=joints)
