"""Golden vectors for the rotation-distance metrics of the tool-pose cost, from the REFERENCE's own
pure-torch functions (curobo/_src/geom/quaternion.py: angular_distance_axis_angle = rotation angle
of the relative rotation, angular_distance_phi3 = acos|<q1,q2>| / (pi/2)), run on CPU:
    PYTHONPATH=/root/reference python tests/golden/make_rotation_metric_golden.py
The Warp tool-pose kernel itself cannot run here; these functions state the same two metrics
(rotation methods 0 and 1).  angular_distance_axis_angle is evaluated per quaternion pair (on a batch
its keepdim norm broadcasts against the scalar part)."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.modules.setdefault("warp", MagicMock())
from curobo._src.geom.quaternion import angular_distance_axis_angle, angular_distance_phi3  # noqa: E402

rng = np.random.default_rng(12)
n = 96
unit = lambda a: (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)  # noqa: E731
cur, goal = unit(rng.standard_normal((n, 4))), unit(rng.standard_normal((n, 4)))
goal[:8] = cur[:8]                      # identical orientations
goal[8:16] = -cur[8:16]                 # the same rotation, opposite sign
small = unit(np.concatenate([np.ones((8, 1)), 1e-3 * rng.standard_normal((8, 3))], axis=1))
w, x, y, z = cur[16:24].T               # goal = cur * small rotation
sw, sx, sy, sz = small.T
goal[16:24] = unit(np.stack([w * sw - x * sx - y * sy - z * sz, w * sx + x * sw + y * sz - z * sy,
                             w * sy - x * sz + y * sw + z * sx, w * sz + x * sy - y * sx + z * sw], axis=1))
angle = np.array([float(angular_distance_axis_angle(torch.tensor(goal[i]), torch.tensor(cur[i]))) for i in range(n)], np.float32)
phi3 = angular_distance_phi3(torch.tensor(goal), torch.tensor(cur)).numpy().astype(np.float32)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rotation_metric_golden.npz")
np.savez_compressed(path, current_quat=cur, goal_quat=goal, axis_angle=angle, phi3=phi3)
print(path, os.path.getsize(path), angle[:3], phi3[:3], angle[16:19])
