"""Robot YAML + URDF -> the flat tensors the kinematics / self-collision kernels read.

Host-side mirror of the reference's model preparation (it is init-path code, not hot-path):
``curobo/_src/robot/loader/kinematics_loader.py:49-175`` (driver), ``:214-262`` (tree order),
``:367-486`` (index tables + chain CSR), ``:685-846`` (locked joints -> fixed transforms),
``:848-916`` (collision spheres) and ``curobo/_src/robot/types/self_collision_params.py:61-236``
(self-collision pair list).  Link/joint *index order* is part of the result contract (the kernels
compose ``cumul[l] = cumul[link_map[l]] @ local[l]`` in index order), so the ordering rules of the
reference are kept; everything else is written for numpy, with no torch / trimesh / yourdfpy.
"""

from __future__ import annotations

import copy
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import yaml

from .urdf import UrdfModel, load_urdf

# joint type codes: reference kernels/kinematics/kinematics_constants.h:10-16
FIXED, X_PRISM, Y_PRISM, Z_PRISM, X_ROT, Y_ROT, Z_ROT = -1, 0, 1, 2, 3, 4, 5
_JOINT_TYPE_BY_NAME = {
    "FIXED": FIXED, "X_PRISM": X_PRISM, "Y_PRISM": Y_PRISM, "Z_PRISM": Z_PRISM,
    "X_ROT": X_ROT, "Y_ROT": Y_ROT, "Z_ROT": Z_ROT,
}


def pose7_to_matrix(p: List[float]) -> np.ndarray:
    """[x y z qw qx qy qz] -> 3x4 (reference LinkParams.create, types/link_params.py:44-60)."""
    x, y, z, qw, qx, qy, qz = [float(v) for v in p]
    n = math.sqrt(qw * qw + qx * qx + qy * qy + qz * qz)
    qw, qx, qy, qz = qw / n, qx / n, qy / n, qz / n
    R = np.array(
        [
            [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
            [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
            [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)],
        ]
    )
    T = np.zeros((3, 4))
    T[:, :3] = R
    T[:, 3] = [x, y, z]
    return T


@dataclass
class _Body:
    link_name: str
    parent_link_name: Optional[str]
    joint_name: str
    joint_type: int
    fixed_transform: np.ndarray  # 3x4
    joint_offset: List[float] = field(default_factory=lambda: [1.0, 0.0])
    mimic_joint_name: Optional[str] = None
    joint_limits: Optional[List[float]] = None
    joint_velocity_limits: List[float] = field(default_factory=lambda: [-2.0, 2.0])
    joint_effort_limit: float = 10000.0
    mass: float = 0.01
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia6: np.ndarray = field(default_factory=lambda: np.array([1e-4, 1e-4, 1e-4, 0, 0, 0.0]))


@dataclass
class RobotModel:
    """Flat numpy mirror of the reference ``KinematicsParams`` (+ self-collision + limits).

    Field names follow ``curobo/_src/robot/types/kinematics_params.py:22-163``.
    """

    fixed_transforms: np.ndarray  # [L,3,4] f32
    link_map: np.ndarray  # [L] i16 parent link index
    joint_map: np.ndarray  # [L] i16 joint index (-1 fixed)
    joint_map_type: np.ndarray  # [L] i8
    joint_offset_map: np.ndarray  # [L*2] f32 (multiplier, offset)
    tool_frame_map: np.ndarray  # [T] i16
    link_sphere_idx_map: np.ndarray  # [S] i16
    link_chain_data: np.ndarray  # CSR i16
    link_chain_offsets: np.ndarray  # [L+1] i16
    joint_links_data: np.ndarray  # CSR i16
    joint_links_offsets: np.ndarray  # [D+1] i16
    joint_affects_endeffector: np.ndarray  # [D*T] bool
    link_spheres: np.ndarray  # [num_envs,S,4] f32
    link_masses_com: np.ndarray  # [L,4] f32
    link_inertias: np.ndarray  # [L,8] f32
    collision_pairs: np.ndarray  # [P,2] i16
    sphere_padding: np.ndarray  # [S] f32
    joint_limits_position: np.ndarray  # [2,D]
    joint_limits_velocity: np.ndarray  # [2,D]
    joint_limits_effort: np.ndarray  # [D]
    num_dof: int
    joint_names: List[str]
    link_names: List[str]
    tool_frames: List[str]
    lock_joints: Dict[str, float]
    cspace: Dict
    base_link: str
    #: links in contact with a grasped object (robot yaml ``grasp_contact_link_names``; the grasp planner switches their
    #: collision spheres off for the final approach and the lift, reference motion_planner.py:437-440)
    grasp_contact_link_names: Optional[List[str]] = None
    #: {actuated joint: [{"joint_name": mimic joint, "joint_offset": [multiplier, offset]}, ...]} (reference
    #: KinematicsParams.mimic_joints, kinematics_loader.py:405-414); None for models saved before the field existed
    mimic_joints: Optional[Dict[str, List[Dict]]] = None

    @property
    def num_links(self) -> int:
        return int(self.link_map.shape[0])

    @property
    def num_spheres(self) -> int:
        return int(self.link_spheres.shape[1])

    _ARRAY_FIELDS = (
        "fixed_transforms", "link_map", "joint_map", "joint_map_type", "joint_offset_map",
        "tool_frame_map", "link_sphere_idx_map", "link_chain_data", "link_chain_offsets",
        "joint_links_data", "joint_links_offsets", "joint_affects_endeffector", "link_spheres",
        "link_masses_com", "link_inertias", "collision_pairs", "sphere_padding",
        "joint_limits_position", "joint_limits_velocity", "joint_limits_effort",
    )

    def as_dict(self) -> Dict[str, np.ndarray]:
        d = {k: getattr(self, k) for k in self._ARRAY_FIELDS}
        d["num_dof"] = np.int64(self.num_dof)
        return d

    def save_npz(self, path: str) -> None:
        meta = dict(
            joint_names=np.array(self.joint_names), link_names=np.array(self.link_names),
            tool_frames=np.array(self.tool_frames), base_link=np.array(self.base_link),
            lock_joint_names=np.array(list(self.lock_joints.keys())),
            lock_joint_values=np.array(list(self.lock_joints.values()), dtype=np.float64),
            cspace_max_acceleration=np.atleast_1d(np.asarray(self.cspace.get("max_acceleration", 10.0), dtype=np.float64)),
            cspace_max_jerk=np.atleast_1d(np.asarray(self.cspace.get("max_jerk", 500.0), dtype=np.float64)),
            cspace_default_joint_position=np.array(
                self.cspace.get("default_joint_position", [0.0] * self.num_dof), dtype=np.float64
            ),
        )
        if self.grasp_contact_link_names is not None:
            meta["grasp_contact_link_names"] = np.array(list(self.grasp_contact_link_names), dtype=str)
        if self.mimic_joints is not None:
            import json

            meta["mimic_joints_json"] = np.array(json.dumps(self.mimic_joints))
        np.savez_compressed(path, **self.as_dict(), **meta)

    @staticmethod
    def load_npz(path: str) -> "RobotModel":
        z = np.load(path, allow_pickle=False)
        kw = {k: z[k] for k in RobotModel._ARRAY_FIELDS}
        lock = dict(zip([str(x) for x in z["lock_joint_names"]], [float(x) for x in z["lock_joint_values"]]))
        return RobotModel(
            **kw,
            num_dof=int(z["num_dof"]),
            joint_names=[str(x) for x in z["joint_names"]],
            link_names=[str(x) for x in z["link_names"]],
            tool_frames=[str(x) for x in z["tool_frames"]],
            lock_joints=lock,
            cspace=dict(
                max_acceleration=[float(x) for x in np.atleast_1d(z["cspace_max_acceleration"])],
                max_jerk=[float(x) for x in np.atleast_1d(z["cspace_max_jerk"])],
                default_joint_position=[float(x) for x in z["cspace_default_joint_position"]],
            ),
            base_link=str(z["base_link"]),
            grasp_contact_link_names=[str(x) for x in z["grasp_contact_link_names"]] if "grasp_contact_link_names" in z.files else None,
            mimic_joints=__import__("json").loads(str(z["mimic_joints_json"])) if "mimic_joints_json" in z.files else None,
        )


class _TreeBuilder:
    def __init__(self, cfg: Dict, urdf: UrdfModel):
        self.cfg = cfg
        self.urdf = urdf
        self.extra_links: Dict[str, Dict] = copy.deepcopy(cfg.get("extra_links") or {})
        # parent map: reference parser_urdf.py:70-78 + parser_base.py:34-49
        self.parent: Dict[str, Dict] = {}
        for jname, j in urdf.joints.items():
            self.parent[j.child] = {"parent": j.parent, "joint_name": jname}
        for name, e in self.extra_links.items():
            self.parent[name] = {"parent": e["parent_link_name"]}
            if e.get("child_link_name") is not None:
                self.parent[e["child_link_name"]]["parent"] = name
        self.bodies: List[_Body] = []
        self.name_to_idx: Dict[str, int] = {}
        self.joint_names: List[str] = []
        self.controlled: List[int] = []

    # -- reference parser_base.py:103-122
    def chain(self, base: str, tip: str) -> List[str]:
        out = [tip]
        link = tip
        while link != base:
            if link not in self.parent:
                raise ValueError(f"link {link!r} has no parent; cannot reach base {base!r}")
            link = self.parent[link]["parent"]
            out.append(link)
        out.reverse()
        return out

    def _limits(self, j):
        """reference parser_urdf.py:90-131"""
        jtype = j.type
        lower, upper = j.lower, j.upper
        if jtype == "continuous":
            jtype, lower, upper = "revolute", -6.28, 6.28
        vel = 100.0 if j.velocity is None else j.velocity
        eff = 100.0 if j.effort is None else j.effort
        return {"lower": lower, "upper": upper, "velocity": vel, "effort": eff}, jtype

    # -- reference parser_urdf.py:133-311 (get_link_parameters)
    def body_for(self, link_name: str, base: bool) -> _Body:
        if link_name in self.extra_links:
            e = self.extra_links[link_name]
            b = _Body(
                link_name=e["link_name"], parent_link_name=e["parent_link_name"],
                joint_name=e["joint_name"], joint_type=_JOINT_TYPE_BY_NAME[e["joint_type"]],
                fixed_transform=pose7_to_matrix(e["fixed_transform"]),
            )
            if e.get("joint_limits") is not None:
                b.joint_limits = [float(x) for x in e["joint_limits"]]
            if e.get("joint_velocity_limits") is not None:
                b.joint_velocity_limits = [float(x) for x in e["joint_velocity_limits"]]
            if e.get("joint_offset") is not None:
                b.joint_offset = [float(x) for x in e["joint_offset"]]
            if e.get("link_mass") is not None:
                b.mass = float(e["link_mass"])
            return b
        link = self.urdf.links[link_name]
        mass, com = 0.01, np.zeros(3)
        inertia6 = np.array([1e-2, 1e-2, 1e-2, 0.0, 0.0, 0.0]) * mass
        if link.mass is not None:
            mass = link.mass if link.mass > 0.0 else 0.01
            # parser_urdf.py:157-170 composes the inertial origin AS A POSE (R, t) with the pose (I, t) and keeps the position of
            # the product: com = R t + t (twice the URDF's offset when the inertial frame is not rotated).  Reproduced as it is:
            # these are the numbers the reference's centre-of-mass output and its RNEA torques are computed from.
            T = link.inertial_origin if link.inertial_origin is not None else np.eye(4)
            com = T[:3, :3] @ T[:3, 3] + T[:3, 3]
            # NOTE: the reference computes an inertia array here but then stores the default
            # (`body_params["link_inertia"] = inertia`, parser_urdf.py:305); kept as is.
        if base:
            return _Body(link_name, None, "base_joint", FIXED, np.eye(4)[:3, :4], mass=mass, com=com,
                         inertia6=inertia6)
        pd = self.parent[link_name]
        j = self.urdf.joints[pd["joint_name"]]
        b = _Body(link_name, pd["parent"], j.name, FIXED, j.origin[:3, :4].copy(), mass=mass, com=com,
                  inertia6=inertia6)
        if j.type == "fixed":
            return b
        lim, jtype = self._limits(j)
        offset = [1.0, 0.0]
        if j.mimic_joint is not None:
            offset = [j.mimic_multiplier, j.mimic_offset]
            b.mimic_joint_name = j.name
            b.joint_name = j.mimic_joint
            lim, _ = self._limits(self.urdf.joints[j.mimic_joint])
        axis = list(j.axis)
        kinds = {"prismatic": (X_PRISM, Y_PRISM, Z_PRISM), "revolute": (X_ROT, Y_ROT, Z_ROT)}
        if jtype not in kinds:
            raise ValueError(f"joint type {jtype!r} not supported")
        code = None
        for a in range(3):
            if abs(axis[a]) == 1:
                code = kinds[jtype][a]
        if code is None:
            raise ValueError(f"joint {j.name}: only axis-aligned joints are supported, got {axis}")
        if -1 in axis:
            offset[0] = -1.0 * offset[0]
        b.joint_type = code
        b.joint_offset = offset
        b.joint_limits = [lim["lower"], lim["upper"]]
        b.joint_velocity_limits = [-lim["velocity"], lim["velocity"]]
        b.joint_effort_limit = lim["effort"]
        return b

    # -- reference kinematics_loader.py:918-958
    def add(self, link_name: str, base: bool = False) -> None:
        idx = len(self.bodies)
        b = self.body_for(link_name, base)
        self.bodies.append(b)
        if b.joint_type != FIXED:
            self.controlled.append(idx)
            if b.joint_name not in self.joint_names:
                self.joint_names.append(b.joint_name)
        self.name_to_idx[b.link_name] = idx

    # -- reference kinematics_loader.py:214-262
    def build_chain(self, base: str, tool_frames: List[str], other_links: List[str]) -> List[str]:
        names = self.chain(base, tool_frames[0])
        self.add(names[0], base=True)
        for n in names[1:]:
            self.add(n)
        for link in other_links:
            if link in self.name_to_idx or link in self.extra_links:
                continue
            for k in self.chain(base, link):
                if k in names:
                    continue
                names.append(k)
                self.add(k)
        for e in self.extra_links:
            if e not in names:
                self.add(e)
                names.append(e)
        return names


def _local_transform(F: np.ndarray, jtype: int, value: float) -> np.ndarray:
    """float64 version of the kernel's local transform (used only to freeze locked joints)."""
    T = np.eye(4)
    T[:3, :4] = F
    J = np.eye(4)
    if jtype in (X_PRISM, Y_PRISM, Z_PRISM):
        J[jtype, 3] = value
    elif jtype in (X_ROT, Y_ROT, Z_ROT):
        a = jtype - X_ROT
        c, s = math.cos(value), math.sin(value)
        i, k = (a + 1) % 3, (a + 2) % 3
        J[i, i], J[i, k], J[k, i], J[k, k] = c, -s, s, c
    return (T @ J)[:3, :4]


def self_collision_pairs(
    collision_link_names: List[str],
    name_to_idx: Dict[str, int],
    ignore: Dict[str, List[str]],
    link_padding: Dict[str, float],
    link_spheres: np.ndarray,
    link_sphere_idx_map: np.ndarray,
):
    """Pair list + per-sphere padding (reference self_collision_params.py:61-236).

    A sphere pair (i<j) is checked iff its two links are different collision links and NEITHER
    link lists the other in ``self_collision_ignore`` (the reference takes the element-wise
    minimum of a directed -inf matrix and its transpose, :214), and sphere ``i`` has at least one
    valid partner.  Pairs are emitted in (i, j) lexicographic order.
    """
    S = link_spheres.shape[0]
    padding = np.zeros(S, dtype=np.float32)
    allowed = np.zeros((S, S), dtype=bool)
    link_padding = dict(link_padding)
    sph_of = {n: np.nonzero(link_sphere_idx_map == name_to_idx[n])[0] for n in collision_link_names}
    for a in collision_link_names:
        padding[sph_of[a]] = link_padding.setdefault(a, 0.0)
        skip = ignore.get(a, [])
        for b in collision_link_names:
            if b == a or b in skip:
                continue
            allowed[np.ix_(sph_of[a], sph_of[b])] = True
    allowed = allowed & allowed.T
    pairs = [(i, j) for i in range(S) for j in range(i + 1, S) if allowed[i, j]]
    return np.asarray(pairs, dtype=np.int16).reshape(-1, 2), padding


def load_robot_model(robot_yaml: str, assets_root: str, num_envs: int = 1) -> RobotModel:
    """Load ``<configs>/robot/*.yml`` + its URDF into a :class:`RobotModel`.

    ``assets_root`` is the directory that contains ``robot/<name>_description`` (the reference's
    ``curobo/content/assets``).
    """
    with open(robot_yaml) as f:
        data = yaml.safe_load(f)
    if "robot_cfg" in data:
        data = data["robot_cfg"]
    cfg = copy.deepcopy(data["kinematics"])
    urdf = load_urdf(os.path.join(assets_root, cfg["urdf_path"]))
    return build_robot_model(cfg, urdf, num_envs=num_envs)


def build_robot_model(cfg: Dict, urdf: UrdfModel, num_envs: int = 1) -> RobotModel:
    base_link = cfg["base_link"]
    tool_frames = list(cfg["tool_frames"])
    collision_link_names = list(cfg.get("collision_link_names") or [])
    collision_spheres = copy.deepcopy(cfg.get("collision_spheres"))
    if isinstance(collision_spheres, str):
        raise ValueError("collision_spheres must be inline in the robot yaml")
    # reference kinematics_loader_cfg.py:169-183
    for k, n in (cfg.get("extra_collision_spheres") or {}).items():
        collision_spheres[k] = [{"center": [0.0, 0.0, 0.0], "radius": -100.0} for _ in range(n)]
    tb = _TreeBuilder(cfg, urdf)
    # reference kinematics_loader.py:97-107
    other_links = list(tool_frames)
    for n in collision_link_names:
        if n not in tool_frames:
            other_links.append(n)
    for e in tb.extra_links.values():
        p = e["parent_link_name"]
        if p not in tool_frames and p not in other_links:
            other_links.append(p)
    chain_names = tb.build_chain(base_link, tool_frames, other_links)
    bodies = tb.bodies
    L = len(bodies)
    joint_names = list(tb.joint_names)

    link_map = np.zeros(L, dtype=np.int16)
    joint_map = np.full(L, -1, dtype=np.int16)
    joint_map_type = np.full(L, -1, dtype=np.int8)
    joint_offset = np.zeros((L, 2), dtype=np.float32)
    joint_offset[0] = [1.0, 0.0]
    fixed = np.stack([b.fixed_transform for b in bodies]).astype(np.float64)
    for i in range(1, L):
        b = bodies[i]
        link_map[i] = tb.name_to_idx[b.parent_link_name]
        joint_offset[i] = b.joint_offset
        joint_map_type[i] = b.joint_type
        if i in tb.controlled:
            joint_map[i] = joint_names.index(b.joint_name)

    # ---- locked joints become fixed transforms (reference kinematics_loader.py:685-846)
    lock_cfg = dict(cfg.get("lock_joints") or {})
    locked: Dict[str, float] = {}
    for jname, value in lock_cfg.items():
        links = [i for i in range(L) if i in tb.controlled and bodies[i].joint_name == jname]
        if not links:
            continue
        for i in links:
            b = bodies[i]
            angle = b.joint_offset[0] * float(value) + b.joint_offset[1]
            fixed[i] = _local_transform(fixed[i], b.joint_type, angle)
            joint_map_type[i] = FIXED
            joint_map[i] = -1
            tb.controlled.remove(i)
        jidx = joint_names.index(jname)
        joint_map[joint_map > jidx] -= 1
        joint_names.remove(jname)
        locked[jname] = float(value)
    D = len(joint_names)

    # tool-frame order: body order when nothing was locked (reference :392-410 reorders to
    # `ordered_link_names`), the configured order otherwise (:776-781 restores it).
    if not locked:
        tool_frames = sorted(tool_frames, key=chain_names.index)
    tool_frame_map = np.asarray([chain_names.index(t) for t in tool_frames], dtype=np.int16)

    # ---- chain CSR: links from base to each link (reference :413-433)
    chain_data: List[int] = []
    chain_offsets = [0]
    for name in chain_names:
        chain_data.extend(tb.name_to_idx[k] for k in tb.chain(base_link, name))
        chain_offsets.append(len(chain_data))
    # ---- joint -> links CSR and joint x tool-frame reachability (reference :285-365)
    jl_data: List[int] = []
    jl_offsets = [0]
    for j in range(D):
        jl_data.extend(int(i) for i in range(L) if joint_map[i] == j)
        jl_offsets.append(len(jl_data))
    affects = np.zeros((D, len(tool_frames)), dtype=bool)
    for t, tname in enumerate(tool_frames):
        ee_chain = set(tb.chain(base_link, tname))
        for j in range(D):
            affects[j, t] = any(chain_names[i] in ee_chain for i in jl_data[jl_offsets[j]:jl_offsets[j + 1]])

    # ---- collision spheres (reference :848-916)
    sph_rows: List[List[float]] = []
    sph_link: List[int] = []
    buf = cfg.get("collision_sphere_buffer", 0.0)
    for name in collision_link_names:
        off = float(buf) if isinstance(buf, (int, float)) else float(buf.get(name, 0.0))
        for s in collision_spheres[name]:
            sph_rows.append([*map(float, s["center"]), float(s["radius"]) + off])
            sph_link.append(tb.name_to_idx[name])
    if sph_rows:
        spheres = np.asarray(sph_rows, dtype=np.float32)
        sphere_link = np.asarray(sph_link, dtype=np.int16)
        pairs, padding = self_collision_pairs(
            collision_link_names, tb.name_to_idx, cfg.get("self_collision_ignore") or {},
            cfg.get("self_collision_buffer") or {}, spheres, sphere_link,
        )
    else:
        spheres = np.zeros((0, 4), np.float32)
        sphere_link = np.zeros((0,), np.int16)
        pairs, padding = np.zeros((0, 2), np.int16), np.zeros((0,), np.float32)

    # ---- limits
    pos_lim = np.zeros((2, D))
    vel_lim = np.zeros((2, D))
    eff_lim = np.zeros((D,))
    for j, jname in enumerate(joint_names):
        b = next(bb for i, bb in enumerate(bodies) if bb.joint_name == jname and bb.mimic_joint_name is None
                 and bb.joint_type != FIXED)
        pos_lim[:, j] = b.joint_limits
        vel_lim[:, j] = b.joint_velocity_limits
        eff_lim[j] = b.joint_effort_limit

    # the cspace block's position clip shrinks the position range and its velocity scale the velocity range
    # (kinematics_loader.py:1102-1124 _update_joint_limits; the scale is per cspace joint name, reindexed to the active joints by
    # CSpaceParams.inplace_reindex, cspace_params.py:149-171; a clip given as a list is applied in the order it is written)
    cs = cfg.get("cspace") or {}

    def per_active_joint(key, value):
        """a scalar / one-element list (every joint) or a list -> [D] in the order of the ACTIVE joints: a list as long as
        cspace.joint_names is reindexed by name (CSpaceParams.inplace_reindex), a list of length D is taken as written"""
        if isinstance(value, (int, float)):
            return np.full(D, float(value))
        v = [float(x) for x in np.asarray(value, np.float64).reshape(-1)]
        if len(v) == 1:
            return np.full(D, v[0])
        names = cs.get("joint_names")
        if names is not None and len(names) == len(v):
            lut = dict(zip(names, v))
            missing = [n for n in joint_names if n not in lut]
            if missing:
                raise ValueError(f"cspace.{key}: cspace.joint_names does not list the active joint(s) {missing}")
            return np.asarray([lut[n] for n in joint_names])
        if len(v) == D:
            return np.asarray(v)
        raise ValueError(f"cspace.{key} holds {len(v)} values: expected one, one per active joint ({D}: {list(joint_names)})"
                         + (f" or one per cspace.joint_names entry ({len(names)})" if names is not None else
                            " (or give cspace.joint_names to reindex a longer list by name)"))

    clip = per_active_joint("position_limit_clip", cs.get("position_limit_clip", 0.0))
    pos_lim[0] += clip
    pos_lim[1] -= clip
    vel_lim = vel_lim * per_active_joint("velocity_scale", cs.get("velocity_scale", 1.0))[None]

    masses_com = np.stack([np.concatenate([b.com, [b.mass]]) for b in bodies]).astype(np.float32)
    inertias = np.zeros((L, 8), dtype=np.float32)
    for i, b in enumerate(bodies):
        inertias[i, :6] = b.inertia6
    cspace = dict(cfg.get("cspace") or {})
    # keep only the active joints of cspace lists (reference CSpaceParams.inplace_reindex)
    if cspace.get("joint_names") is not None and cspace.get("default_joint_position") is not None:
        lut = dict(zip(cspace["joint_names"], cspace["default_joint_position"]))
        cspace["default_joint_position"] = [float(lut.get(n, 0.0)) for n in joint_names]
    # acceleration / jerk limits per ACTIVE joint (reference: JointLimits.acceleration / .jerk = -+ CSpaceParams.max_acceleration /
    # max_jerk after inplace_reindex, kinematics_loader.py:1102-1112; scalars are broadcast, cspace_params.py:45-50, 84-110)
    for key, default in (("max_acceleration", 10.0), ("max_jerk", 500.0)):
        cspace[key] = [float(x) for x in per_active_joint(key, cspace.get(key, default))]

    mimic: Dict[str, List[Dict]] = {}
    for b in bodies[1:]:
        if b.mimic_joint_name is not None:  # (followers of locked joints included: the reference fills the table before it locks them)
            mimic.setdefault(b.joint_name, []).append({"joint_name": b.mimic_joint_name, "joint_offset": [float(b.joint_offset[0]), float(b.joint_offset[1])]})

    return RobotModel(
        fixed_transforms=fixed.astype(np.float32),
        link_map=link_map, joint_map=joint_map, joint_map_type=joint_map_type,
        joint_offset_map=joint_offset.reshape(-1).copy(),
        tool_frame_map=tool_frame_map,
        link_sphere_idx_map=sphere_link,
        link_chain_data=np.asarray(chain_data, dtype=np.int16),
        link_chain_offsets=np.asarray(chain_offsets, dtype=np.int16),
        joint_links_data=np.asarray(jl_data, dtype=np.int16),
        joint_links_offsets=np.asarray(jl_offsets, dtype=np.int16),
        joint_affects_endeffector=affects.reshape(-1).copy(),
        link_spheres=np.repeat(spheres[None], num_envs, axis=0),
        link_masses_com=masses_com, link_inertias=inertias,
        collision_pairs=pairs, sphere_padding=padding,
        joint_limits_position=pos_lim, joint_limits_velocity=vel_lim, joint_limits_effort=eff_lim,
        num_dof=D, joint_names=joint_names, link_names=list(chain_names),
        tool_frames=tool_frames, lock_joints=locked, cspace=cspace, base_link=base_link,
        grasp_contact_link_names=list(cfg["grasp_contact_link_names"]) if cfg.get("grasp_contact_link_names") else None,
        mimic_joints=mimic,
    )
