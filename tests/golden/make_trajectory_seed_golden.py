"""Golden vectors for the trajectory seed generator from the REFERENCE's own class
(curobo/_src/util/trajectory_seed_generator.py, pure torch, CPU):
    PYTHONPATH=/root/reference python tests/golden/make_trajectory_seed_golden.py
(`warp` is absent here and unused by this class: a stub module stands in for it during the import)."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.modules.setdefault("warp", MagicMock())
from curobo._src.types.device_cfg import DeviceCfg  # noqa: E402
from curobo._src.util.trajectory_seed_generator import TrajectorySeedGenerator  # noqa: E402

torch.manual_seed(7)
B, S, H, D = 3, 5, 12, 7
gen = TrajectorySeedGenerator(H, D, DeviceCfg(device=torch.device("cpu")))
start, goal = torch.randn(B, D), torch.randn(B, S, D)
out = {"start": start, "goal": goal, "interpolated": gen.generate_interpolated_seeds(start, goal, S),
       "constant": gen.generate_constant_seeds(start, S)}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trajectory_seed_golden.npz")
np.savez_compressed(path, **{k: v.numpy() for k, v in out.items()})
print(path, os.path.getsize(path), out["interpolated"].shape)
