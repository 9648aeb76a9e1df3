"""``curobo_amd.types.Pose.multiply / inverse / from_matrix`` against the reference's ``Pose`` (types/pose.py; its Warp kernels run through the
stand-in) on random poses.   python tests/golden/compare_pose_ops.py        (needs /root/reference)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401
import torch  # noqa: E402
from curobo._src.types.pose import Pose as Ref  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.types import Pose as Ours  # noqa: E402

rng = np.random.default_rng(6)
n = 40
p1, p2 = rng.uniform(-2, 2, (n, 3)).astype(np.float32), rng.uniform(-2, 2, (n, 3)).astype(np.float32)
q1, q2 = rng.normal(size=(n, 4)).astype(np.float32), rng.normal(size=(n, 4)).astype(np.float32)
q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
t = torch.as_tensor
ra, rb = Ref(position=t(p1), quaternion=t(q1)), Ref(position=t(p2), quaternion=t(q2))
oa, ob = Ours(position=t(p1), quaternion=t(q1)), Ours(position=t(p2), quaternion=t(q2))
ok = True


def same_rotation(a, b):  # q and -q are one rotation
    a, b = a.detach().cpu().numpy().reshape(-1, 4), b.detach().cpu().numpy().reshape(-1, 4)
    return np.abs(np.abs((a * b).sum(-1)) - 1.0).max() < 2e-6


for what, r, o in (("multiply", ra.multiply(rb), oa.multiply(ob)), ("inverse", ra.inverse(), oa.inverse()),
                   ("inverse o multiply", ra.inverse().multiply(rb), oa.inverse().multiply(ob))):
    dp = float(np.abs(r.position.detach().cpu().numpy().reshape(-1, 3) - o.position.detach().cpu().numpy().reshape(-1, 3)).max())
    good = dp < 5e-6 and same_rotation(r.quaternion, o.quaternion)
    ok &= good
    print(f"{what}: {'ok' if good else 'DIFFERENT'} (max |dp| {dp:.2e})")
# Pose.from_matrix (reference: Warp's quat_from_matrix through the stand-in): rotations incl. half turns about the axes
from scipy.spatial.transform import Rotation  # noqa: E402

mats = np.tile(np.eye(4, dtype=np.float32), (n + 3, 1, 1))
mats[:n, :3, :3] = Rotation.from_quat(q1[:, [1, 2, 3, 0]]).as_matrix()
mats[n:, :3, :3] = Rotation.from_rotvec(np.pi * np.eye(3)).as_matrix()
mats[:n, :3, 3] = p1
r, o = Ref.from_matrix(t(mats)), Ours.from_matrix(t(mats))
dp = float(np.abs(r.position.detach().cpu().numpy().reshape(-1, 3) - o.position.detach().cpu().numpy().reshape(-1, 3)).max())
good = dp < 1e-6 and same_rotation(r.quaternion, o.quaternion)
ok &= good
print(f"from_matrix: {'ok' if good else 'DIFFERENT'} (max |dp| {dp:.2e})")
# the convenience members (types/pose.py:117-670): matrices, Euler constructors, distances, point transforms, shape helpers
def close(a, b, tol=2e-6):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max()) < tol


from curobo._src.types.device_cfg import DeviceCfg as RefDeviceCfg  # noqa: E402

cpu_cfg = RefDeviceCfg(device=torch.device("cpu"))
pts = t(rng.normal(size=(n, 7, 3)).astype(np.float32))
eul = t(rng.uniform(-3, 3, (n, 3)).astype(np.float32))
checks = {
    "get_matrix": close(ra.get_matrix(), oa.get_matrix()),
    "get_affine_matrix": close(ra.get_affine_matrix(), oa.get_affine_matrix()),
    "get_rotation": close(ra.get_rotation(), oa.get_rotation()),
    "get_pose_vector": close(ra.get_pose_vector(), oa.get_pose_vector()),
    "linear_distance": close(ra.linear_distance(rb), oa.linear_distance(ob)),
    # (the reference's axis-angle form returns [n, n] for n > 1 -- atan2 of a kept dimension against a dropped one broadcasts;
    # its docstring says [...], which is its diagonal and what this package returns)
    "angular_distance": close(torch.diagonal(ra.angular_distance(rb)), oa.angular_distance(ob), 5e-6)
    and close(ra[2:3].angular_distance(rb[2:3]).reshape(-1), oa[2:3].angular_distance(ob[2:3]), 5e-6),
    "angular_distance phi3": close(ra.angular_distance(rb, use_phi3=True), oa.angular_distance(ob, use_phi3=True), 5e-6),
    "from_euler_xyz": same_rotation(Ref.from_euler_xyz(eul).quaternion, Ours.from_euler_xyz(eul).quaternion),
    "from_euler_xyz_intrinsic": same_rotation(Ref.from_euler_xyz_intrinsic(eul).quaternion, Ours.from_euler_xyz_intrinsic(eul).quaternion),
    "batch_transform_points": close(ra.batch_transform_points(pts), oa.batch_transform_points(pts), 5e-6),
    "batch_transform_points_inverse": close(ra.batch_transform_points_inverse(pts), oa.batch_transform_points_inverse(pts), 5e-6),
    "transform_points": close(ra[3:4].transform_points(pts[0]), oa[3:4].transform_points(pts[0]), 5e-6),
    "compute_local_pose": close(ra.compute_local_pose(rb).position, oa.compute_local_pose(ob).position, 5e-6),
    "compute_offset_pose": close(ra.compute_offset_pose(rb).position, oa.compute_offset_pose(ob).position, 5e-6),
    "repeat_seeds": close(ra.repeat_seeds(3).position, oa.repeat_seeds(3).position) and ra.repeat_seeds(3).position.shape == oa.repeat_seeds(3).position.shape,
    "repeat": ra.repeat(2).quaternion.shape == oa.repeat(2).quaternion.shape and close(ra.repeat(2).quaternion, oa.repeat(2).quaternion),
    "stack / cat": close(ra.stack(rb).position, oa.stack(ob).position) and close(Ref.cat([ra, rb]).quaternion, Ours.cat([oa, ob]).quaternion),
    "get_index / getitem / len": close(ra.get_index(2).position, oa.get_index(2).position) and close(ra[1:4].quaternion, oa[1:4].quaternion) and len(ra) == len(oa),
    "tolist": np.allclose(ra[0:1].tolist(), oa[0:1].tolist(), atol=1e-6) and np.allclose(ra[0:1].tolist(q_xyzw=True), oa[0:1].tolist(q_xyzw=True), atol=1e-6),
    "from_list / from_batch_list": close(Ref.from_list([1, 2, 3, 0, 1, 0, 0], device_cfg=cpu_cfg, q_xyzw=False).quaternion, Ours.from_list([1, 2, 3, 0, 1, 0, 0]).quaternion)
    and same_rotation(Ref.from_batch_list([[1, 2, 3, 0, 0, 0.6, 0.8]], device_cfg=cpu_cfg, q_xyzw=True).quaternion, Ours.from_batch_list([[1, 2, 3, 0, 0, 0.6, 0.8]], q_xyzw=True).quaternion),
    "shape / ndim / device": tuple(ra.shape) == tuple(oa.shape) and ra.ndim == oa.ndim and str(ra.device) == str(oa.device),
}
for k, v in checks.items():
    print(f"member {k}: {'ok' if v else 'DIFFERENT'}")
    ok &= bool(v)
sys.exit(0 if ok else 1)
