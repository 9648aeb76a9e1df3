"""Golden vectors for trajectory retiming from the REFERENCE's own pure-torch helpers
(curobo/_src/util/trajectory.py: calculate_dt_no_clamp, calculate_traj_steps), run on CPU:
    PYTHONPATH=/root/reference python tests/golden/make_retime_golden.py
The module also imports NVIDIA Warp kernels (absent here, unused by these two functions): a stub
module stands in for `warp` during the import."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.modules.setdefault("warp", MagicMock())
from curobo._src.util import trajectory as R  # noqa: E402

torch.manual_seed(0)
v, a, j = torch.randn(5, 33, 7), torch.randn(5, 33, 7) * 5, torch.randn(5, 33, 7) * 50
mv, ma, mj = torch.rand(7) + 1, torch.rand(7) * 5 + 5, torch.rand(7) * 100 + 100
score = R.calculate_dt_no_clamp(v, a, j, mv, ma, mj, epsilon=1e-3)
dt = torch.rand(6) * 0.1 + 0.01
idt = torch.full((6,), 0.02)
out = {"vel": v, "acc": a, "jerk": j, "max_vel": mv, "max_acc": ma, "max_jerk": mj, "score": score, "dt": dt, "idt": idt}
for ni in (False, True):
    steps, smax = R.calculate_traj_steps(dt, idt, 17, nearest_int=ni)
    out[f"steps_{int(ni)}"] = steps
    out[f"steps_max_{int(ni)}"] = smax
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "retime_golden.npz")
np.savez_compressed(path, **{k: np.asarray(t) for k, t in out.items()})
print(path, os.path.getsize(path), score)
