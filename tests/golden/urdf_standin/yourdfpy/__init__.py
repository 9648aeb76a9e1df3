"""Stand-in for the part of ``yourdfpy`` the reference's ``UrdfRobotParser`` reads (see ../README.md): ``URDF.load`` ->
``joint_map`` / ``joint_names`` / ``link_map`` with Joint(name, type, parent, child, origin, axis, limit, mimic) and
Link(name, inertial, visuals, collisions).  Defaults as yourdfpy has them: a missing <origin> is None, a missing <axis> is
(1, 0, 0), a missing <limit> is None, missing limit attributes are None, mimic multiplier 1 / offset 0.  Geometry is not
loaded (visuals / collisions are empty lists): the comparison is about the kinematic tree."""
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


def _floats(text, n, default):
    if text is None:
        return np.asarray(default, np.float64)
    v = np.asarray([float(x) for x in text.split()], np.float64)
    assert v.size == n, text
    return v


def _origin(el) -> Optional[np.ndarray]:
    if el is None:
        return None
    x, y, z = _floats(el.get("xyz"), 3, [0, 0, 0])
    r, p, yw = _floats(el.get("rpy"), 3, [0, 0, 0])
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(yw), np.sin(yw)
    m = np.eye(4)
    # fixed-axis roll (x), pitch (y), yaw (z): R = Rz(yaw) Ry(pitch) Rx(roll)
    m[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    m[:3, 3] = [x, y, z]
    return m


def _opt_float(el, key):
    v = el.get(key)
    return None if v is None else float(v)


@dataclass
class Limit:
    effort: Optional[float] = None
    velocity: Optional[float] = None
    lower: Optional[float] = None
    upper: Optional[float] = None


@dataclass
class Mimic:
    joint: str
    multiplier: float = 1.0
    offset: float = 0.0


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    origin: Optional[np.ndarray] = None
    axis: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0]))
    limit: Optional[Limit] = None
    mimic: Optional[Mimic] = None


@dataclass
class Inertial:
    origin: Optional[np.ndarray] = None
    mass: Optional[float] = None
    inertia: Optional[np.ndarray] = None


@dataclass
class Link:
    name: str
    inertial: Optional[Inertial] = None
    visuals: List = field(default_factory=list)
    collisions: List = field(default_factory=list)


class URDF:
    def __init__(self, links: List[Link], joints: List[Joint]):
        self.link_map: Dict[str, Link] = {l.name: l for l in links}
        self.joint_map: Dict[str, Joint] = {j.name: j for j in joints}
        self.joint_names: List[str] = [j.name for j in joints]
        self.actuated_joint_names: List[str] = [j.name for j in joints if j.type != "fixed" and j.mimic is None]
        children = {j.child for j in joints}
        self.base_link: Optional[str] = next((l.name for l in links if l.name not in children), None)  # root of the tree

    @staticmethod
    def load(fname_or_file, load_meshes=False, build_scene_graph=False, filename_handler=None, **kwargs) -> "URDF":
        root = ET.parse(fname_or_file).getroot()
        links, joints = [], []
        for el in root.findall("link"):
            inertial = None
            ie = el.find("inertial")
            if ie is not None:
                me, te = ie.find("mass"), ie.find("inertia")
                inertia = None
                if te is not None:
                    g = lambda k: float(te.get(k, 0.0))  # noqa: E731
                    inertia = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
                inertial = Inertial(origin=_origin(ie.find("origin")), mass=None if me is None else float(me.get("value")), inertia=inertia)
            links.append(Link(name=el.get("name"), inertial=inertial))
        for el in root.findall("joint"):
            j = Joint(name=el.get("name"), type=el.get("type"), parent=el.find("parent").get("link"), child=el.find("child").get("link"),
                      origin=_origin(el.find("origin")))
            ae = el.find("axis")
            if ae is not None:
                j.axis = _floats(ae.get("xyz"), 3, [1, 0, 0])
            le = el.find("limit")
            if le is not None:
                j.limit = Limit(effort=_opt_float(le, "effort"), velocity=_opt_float(le, "velocity"), lower=_opt_float(le, "lower"),
                                upper=_opt_float(le, "upper"))
            me = el.find("mimic")
            if me is not None:
                j.mimic = Mimic(joint=me.get("joint"), multiplier=float(me.get("multiplier", 1.0)), offset=float(me.get("offset", 0.0)))
            joints.append(j)
        return URDF(links, joints)
