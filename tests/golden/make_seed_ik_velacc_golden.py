"""Golden vectors for the velocity / acceleration regularisation blocks of the seed-IK error, produced by the
REFERENCE's own methods run on CPU (curobo/_src/solver/seed_ik/seed_ik_error_calculator.py:
_compute_velocity_errors :389-419, _compute_acceleration_errors :423-456, pure torch):
    PYTHONPATH=/root/reference python tests/golden/make_seed_ik_velacc_golden.py
`warp` (absent here, unused by these methods) is stubbed for the import, as in make_seed_ik_limits_golden.py."""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.modules.setdefault("warp", MagicMock())
from curobo._src.solver.seed_ik.seed_ik_error_calculator import SeedIKErrorCalculator  # noqa: E402

rng = np.random.default_rng(77)
n, D = 64, 7
q = rng.standard_normal((n, D)).astype(np.float32)
cur = (q + 0.2 * rng.standard_normal((n, D))).astype(np.float32)
vel = (1.5 * rng.standard_normal((n, D))).astype(np.float32)
dt = (0.01 + 0.2 * rng.random(n)).astype(np.float32)
wv, wa = np.float32(0.8), np.float32(0.05)
me = types.SimpleNamespace(config=types.SimpleNamespace(velocity_weight=float(wv), acceleration_weight=float(wa)))
jv, Jv, ev = SeedIKErrorCalculator._compute_velocity_errors(me, torch.tensor(q), torch.tensor(cur), torch.tensor(dt), n)
ja, Ja, ea = SeedIKErrorCalculator._compute_acceleration_errors(me, torch.tensor(q), torch.tensor(cur), torch.tensor(vel), torch.tensor(dt), n)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seed_ik_velacc_golden.npz")
np.savez_compressed(path, q=q, current_position=cur, current_velocity=vel, dt=dt, velocity_weight=wv, acceleration_weight=wa,
                    vel_jTerror=jv.numpy(), vel_jacobian_diag=torch.diagonal(Jv, dim1=1, dim2=2).numpy(), vel_error=ev.numpy(),
                    acc_jTerror=ja.numpy(), acc_jacobian_diag=torch.diagonal(Ja, dim1=1, dim2=2).numpy(), acc_error=ea.numpy())
print(path, os.path.getsize(path))
