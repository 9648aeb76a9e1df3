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
sys.exit(0 if ok else 1)
