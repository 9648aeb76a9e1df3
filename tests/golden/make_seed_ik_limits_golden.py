"""Golden vectors for the joint-limit block of the seed-IK error (with and without velocity
clamping of the bounds), produced by the REFERENCE's own method run on CPU
(curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:_compute_joint_limit_errors, pure torch):
    PYTHONPATH=/root/reference python tests/golden/make_seed_ik_limits_golden.py
The module imports NVIDIA Warp cost kernels (absent here, unused by this method): a stub module
stands in for `warp` during the import; the method is called on a namespace that carries exactly
the attributes it reads."""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.modules.setdefault("warp", MagicMock())
from curobo._src.solver.seed_ik.seed_ik_error_calculator import SeedIKErrorCalculator  # noqa: E402

rng = np.random.default_rng(31)
n, D = 96, 7
lo = -(1.0 + rng.random(D)).astype(np.float32)
hi = (1.0 + rng.random(D)).astype(np.float32)
vlim = np.stack([-(0.5 + 2 * rng.random(D)), 0.5 + 2 * rng.random(D)]).astype(np.float32)
q = (2.6 * rng.standard_normal((n, D))).astype(np.float32) * 0.6
cur = (0.8 * rng.standard_normal((n, D))).astype(np.float32)
dt = (0.02 + 0.3 * rng.random(n)).astype(np.float32)
me = types.SimpleNamespace(action_min=torch.tensor(lo), action_max=torch.tensor(hi), velocity_limits=torch.tensor(vlim),
                           config=types.SimpleNamespace(joint_limit_weight=1.7))
out = {}
for name, active in (("plain", False), ("clamped", True)):
    jte, jac, err = SeedIKErrorCalculator._compute_joint_limit_errors(
        me, torch.tensor(q), n, current_position=torch.tensor(cur), dt=torch.tensor(dt), velocity_clamping_active=active)
    out[f"{name}/jTerror"], out[f"{name}/jacobian"], out[f"{name}/error"] = jte.numpy(), jac.numpy(), err.numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seed_ik_limits_golden.npz")
np.savez_compressed(path, lo=lo, hi=hi, velocity_limits=vlim, q=q, current_position=cur, dt=dt, weight=np.float32(1.7), **out)
print(path, os.path.getsize(path), "violations plain", int((out["plain/jTerror"] != 0).sum()), "clamped",
      int((out["clamped/jTerror"] != 0).sum()))
