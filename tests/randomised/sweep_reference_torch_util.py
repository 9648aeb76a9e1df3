"""The host-side torch utilities of the path (curobo_amd/util: retiming score, interpolation step counts, knot seeds) against the
reference's own torch code (curobo/_src/util/trajectory.py, trajectory_seed_generator.py) on random shapes.  CPU only.
    python tests/randomised/sweep_reference_torch_util.py [cases] [seed]"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/util"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
import torch  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")  # (`curobo` = the reference's package in this process; the product is `curobo_amd`)
sys.modules.setdefault("warp", MagicMock())  # the reference module imports Warp kernels these functions do not use
from curobo._src.types.device_cfg import DeviceCfg  # noqa: E402
from curobo._src.util import trajectory as R  # noqa: E402
from curobo._src.util.trajectory_seed_generator import TrajectorySeedGenerator as RefSeeds  # noqa: E402

from curobo_amd.util.knot_seeds import TrajectorySeedGenerator  # noqa: E402
from curobo_amd.util.trajectory import calculate_dt_no_clamp, calculate_traj_steps  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    torch.manual_seed(int(rng.integers(1 << 30)))
    b, h, d = int(rng.integers(1, 9)), int(rng.choice([2, 17, 33, 65])), int(rng.integers(1, 12))
    try:
        v, a, j = torch.randn(b, h, d) * float(rng.choice([0.01, 1.0, 10.0])), torch.randn(b, h, d) * 5, torch.randn(b, h, d) * 50
        if rng.random() < 0.3:
            v[0] = 0  # a trajectory at rest
        mv, ma, mj = torch.rand(d) + 1, torch.rand(d) * 5 + 5, torch.rand(d) * 100 + 100
        eps = float(rng.choice([1e-3, 1e-2]))
        np.testing.assert_allclose(calculate_dt_no_clamp(v, a, j, mv, ma, mj, epsilon=eps).numpy(), R.calculate_dt_no_clamp(v, a, j, mv, ma, mj, epsilon=eps).numpy(),
                                   rtol=1e-6, err_msg="calculate_dt_no_clamp")
        n = int(rng.integers(1, 12))
        dt, idt = torch.rand(n) * 0.2 + 0.005, torch.full((n,), float(rng.choice([0.01, 0.02, 0.05])))
        hz = int(rng.choice([5, 17, 33, 65]))
        for ni in (False, True):
            s0, m0 = calculate_traj_steps(dt, idt, hz, nearest_int=ni)
            s1, m1 = R.calculate_traj_steps(dt, idt, hz, nearest_int=ni)
            assert np.array_equal(s0.numpy(), s1.numpy()) and int(m0) == int(m1), f"calculate_traj_steps (nearest_int {ni})"
        B, S, H, D = int(rng.integers(1, 6)), int(rng.integers(1, 9)), int(rng.choice([2, 5, 12, 32])), int(rng.integers(1, 10))
        start, goal = torch.randn(B, D), torch.randn(B, S, D)
        ours, theirs = TrajectorySeedGenerator(H, D), RefSeeds(H, D, DeviceCfg(device=torch.device("cpu")))
        assert np.array_equal(ours.generate_interpolated_seeds(start, goal, S).numpy(), theirs.generate_interpolated_seeds(start, goal, S).numpy()), "interpolated seeds"
        assert np.array_equal(ours.generate_constant_seeds(start, S).numpy(), theirs.generate_constant_seeds(start, S).numpy()), "constant seeds"
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
