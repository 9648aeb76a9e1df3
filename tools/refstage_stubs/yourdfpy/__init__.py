"""Stand-in for the third-party ``yourdfpy`` package (absent in this image; the reference pins yourdfpy >= 0.0.53,
pyproject.toml), for running the REFERENCE's own robot loader on the GPU box (tools/reference_on_hip.py).  TEST
INFRASTRUCTURE ONLY.  It offers exactly the members ``curobo/_src/robot/parser/parser_urdf.py`` touches -- ``URDF.load``,
``joint_map`` / ``joint_names`` / ``link_map`` / ``base_link``, ``Joint`` (name, type, parent, child, origin 4x4, axis,
limit, mimic) and ``Link`` (inertial: mass, origin 4x4, inertia 3x3; visuals, collisions) -- read with this repository's
xml.etree URDF reader (``curobo_amd/robot/urdf.py``)."""
from types import SimpleNamespace

import numpy as np

from curobo_amd.robot.urdf import load_urdf


class Joint(SimpleNamespace):
    pass


class Link(SimpleNamespace):
    pass


class URDF:
    def __init__(self, model, path):
        self._model, self._path = model, path
        self.joint_map, self.link_map = {}, {}
        children = set()
        for name, j in model.joints.items():
            has_limit = j.type not in ("fixed",)
            limit = SimpleNamespace(effort=j.effort, lower=j.lower if j.lower is not None else 0.0,
                                    upper=j.upper if j.upper is not None else 0.0, velocity=j.velocity) if has_limit else None
            mimic = (SimpleNamespace(joint=j.mimic_joint, multiplier=j.mimic_multiplier, offset=j.mimic_offset)
                     if j.mimic_joint is not None else None)
            self.joint_map[name] = Joint(name=name, type=j.type, parent=j.parent, child=j.child, origin=np.asarray(j.origin, np.float64),
                                         axis=np.asarray(j.axis, np.float64), limit=limit, mimic=mimic)
            children.add(j.child)
        for name, l in model.links.items():
            inertial = None
            if l.mass is not None:
                inertial = SimpleNamespace(mass=l.mass, origin=l.inertial_origin, inertia=l.inertia if l.inertia is not None else np.zeros((3, 3)))
            self.link_map[name] = Link(name=name, inertial=inertial, visuals=[], collisions=[])
        self.joint_names = list(self.joint_map.keys())
        roots = [n for n in self.link_map if n not in children]
        self.base_link = roots[0] if roots else next(iter(self.link_map))
        self.robot = SimpleNamespace(name="robot", joints=list(self.joint_map.values()), links=list(self.link_map.values()))

    @staticmethod
    def load(fname, load_meshes=False, build_scene_graph=True, filename_handler=None, **kw):
        return URDF(load_urdf(fname), fname)

    def write_xml(self):
        import xml.etree.ElementTree as ET

        return ET.parse(self._path).getroot()
