"""The REFERENCE'S OWN robot loader on the CPU: ``UrdfRobotParser`` + ``KinematicsLoader`` (curobo/_src/robot/parser/parser_urdf.py,
curobo/_src/robot/loader/kinematics_loader.py), unmodified, building the kernel tensors of a robot YAML + URDF -- the data that
feeds every kernel of the path.  Test infrastructure (needs /root/reference); used by tests/test_reference_robot_loader.py.

What stands in for what this image lacks:
  * ``yourdfpy`` / ``lxml``: tests/golden/urdf_standin (URDF read with xml.etree; yourdfpy's defaults);
  * ``warp``: tests/golden/warp_emulator (Pose.from_matrix / Pose.multiply are Warp kernels);
  * a GPU: ``DeviceCfg`` is held to the CPU, the parser's literal ``device="cuda"`` likewise;
  * ``KinematicsLoader._get_link_poses`` (the poses of the links behind locked joints, computed by the reference with its FK
    CUDA kernel on a GPU tensor): the same launch through ``oracle/_ref`` -- the reference's FK kernel compiled for the CPU
    (oracle/cuda_on_cpu) -- on the tensors the reference's loader has built up to that point.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CUROBO_REFERENCE", "/root/reference")
for p in (HERE, os.path.join(HERE, "urdf_standin")):
    if p not in sys.path:
        sys.path.insert(0, p)
if REF not in sys.path:
    sys.path.insert(0, REF)  # (ahead of the repository root: the repository has a ``curobo`` facade of its own)
import make_scene_warp_golden as _emu  # noqa: E402,F401  (the Warp stand-in + stubs for trimesh)
import torch  # noqa: E402

import curobo._src.types.device_cfg as _DC  # noqa: E402


def _cpu_post_init(self):
    object.__setattr__(self, "device", torch.device("cpu"))


_DC.DeviceCfg.__post_init__ = _cpu_post_init
_DC.DeviceCfg.__dataclass_fields__["device"].default = torch.device("cpu")
_DC.DeviceCfg.__init__.__defaults__ = (torch.device("cpu"),) + tuple(_DC.DeviceCfg.__init__.__defaults__[1:])

import curobo._src.robot.parser.parser_urdf as _PU  # noqa: E402
from curobo._src.robot.loader.kinematics_loader import KinematicsLoader  # noqa: E402
from curobo._src.robot.loader.kinematics_loader_cfg import KinematicsLoaderCfg  # noqa: E402
from curobo._src.types.pose import Pose  # noqa: E402
from curobo._src.util_file import load_yaml  # noqa: E402


class _TorchOnCpu:
    """``torch`` for the parser module: ``as_tensor(..., device="cuda")`` lands on the CPU"""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def as_tensor(x, device=None, dtype=None):
        return torch.as_tensor(x, device="cpu", dtype=dtype)


_PU.torch = _TorchOnCpu()


class CpuKinematicsLoader(KinematicsLoader):
    def _get_link_poses(self, q, query_link_names, kinematics_config):
        """the reference launches ``KinematicsFusedFunction`` (its FK kernel) at the lock-joint values and reads the poses of the
        tool frames; here the same kernel runs through oracle/_ref on the reference's tensors"""
        sys.path.insert(0, ROOT)
        from oracle import ref_kernels

        kc = kinematics_config
        n = lambda t, dt: np.ascontiguousarray(t.detach().cpu().numpy().astype(dt))  # noqa: E731
        L = kc.fixed_transforms.shape[0]
        md = dict(fixed_transforms=n(kc.fixed_transforms, np.float32).reshape(L, 3, 4), link_map=n(kc.link_map, np.int16),
                  joint_map=n(kc.joint_map, np.int16), joint_map_type=n(kc.joint_map_type, np.int8),
                  joint_offset_map=n(kc.joint_offset_map, np.float32).reshape(-1), tool_frame_map=n(kc.tool_frame_map, np.int16),
                  link_sphere_idx_map=np.zeros(1, np.int16), link_spheres=np.zeros((1, 1, 4), np.float32),
                  link_masses_com=np.zeros((L, 4), np.float32), num_dof=int(q.numel()),
                  # (tables of the VJP / Jacobian launches: not read by the pose-only forward launch)
                  link_chain_data=np.zeros(1, np.int16), link_chain_offsets=np.zeros(L + 1, np.int16), joint_links_data=np.zeros(1, np.int16),
                  joint_links_offsets=np.zeros(int(q.numel()) + 1, np.int16), joint_affects_endeffector=np.zeros(1, np.bool_))
        out = ref_kernels.ReferenceKernels().kinematics_forward(n(q, np.float32).reshape(1, -1), md, compute_spheres=False)
        pos, quat = out["link_pos"].reshape(-1, 3), out["link_quat"].reshape(-1, 4)
        idx = [self.tool_frames.index(name) for name in query_link_names]
        return Pose(position=torch.as_tensor(pos[idx]).view(1, -1, 3).clone(), quaternion=torch.as_tensor(quat[idx]).view(1, -1, 4).clone())


def reference_kinematics(robot_yaml: str):
    """(KinematicsParams, SelfCollisionKinematicsCfg or None) exactly as the reference's loader builds them from a robot YAML"""
    cfg = load_yaml(robot_yaml)
    cfg = cfg.get("robot_cfg", cfg)["kinematics"]
    loader = CpuKinematicsLoader(KinematicsLoaderCfg(**cfg, device_cfg=_DC.DeviceCfg(device="cpu")))
    return loader.kinematics_config, loader.self_collision_config


def reference_kinematics_from_dict(kinematics_dict: dict):
    """the same from the ``kinematics`` section of a robot configuration given as a dictionary"""
    import copy

    loader = CpuKinematicsLoader(KinematicsLoaderCfg(**copy.deepcopy(kinematics_dict), device_cfg=_DC.DeviceCfg(device="cpu")))
    return loader.kinematics_config, loader.self_collision_config


if __name__ == "__main__":
    kc, sc = reference_kinematics(os.path.join(REF, "curobo", "content", "configs", "robot", f"{sys.argv[1]}.yml"))
    print(kc.fixed_transforms.shape, kc.joint_names, kc.tool_frames)
