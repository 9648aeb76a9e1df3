"""``curobo_amd.util.deceleration`` against the reference's ``TrajectorySeedGenerator.generate_deceleration_seeds``
(util/trajectory_seed_generator.py:122-376) on random states, the three profiles.
    python tests/golden/compare_deceleration_seeds.py        (needs /root/reference)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401
import torch  # noqa: E402
from curobo._src.state.state_joint import JointState as RefJS  # noqa: E402
from curobo._src.types.device_cfg import DeviceCfg as RefCfg  # noqa: E402
from curobo._src.util.trajectory_seed_generator import TrajectorySeedGenerator  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.util.deceleration import deceleration_knots  # noqa: E402

rng = np.random.default_rng(4)
ok = True
cpu = RefCfg(device=torch.device("cpu"))
for n, D, B in ((12, 7, 5), (16, 6, 3), (4, 3, 2)):
    gen = TrajectorySeedGenerator(n, D, cpu)
    p = torch.as_tensor(rng.uniform(-1, 1, (B, D)).astype(np.float32))
    v = torch.as_tensor(rng.uniform(-2, 2, (B, D)).astype(np.float32))
    v[0, 0], v[-1, -1] = 0.0, 5e-7  # joints at rest
    a = torch.as_tensor(rng.uniform(-15, 15, (B, D)).astype(np.float32))
    for profile in ("linear", "exponential", "smooth", "something else"):
        for dt in (0.02, 0.1):
            js = RefJS(position=p.clone(), velocity=v.clone(), acceleration=a.clone(), dt=torch.full((B,), dt))
            ref = gen.generate_deceleration_seeds(js, 2, deceleration_profile=profile)  # [B, 2, n, D]
            ours = deceleration_knots(p, v, a, dt, n, profile)
            good = tuple(ref.shape) == (B, 2, n, D) and torch.equal(ref[:, 0], ref[:, 1]) and float((ref[:, 0] - ours).abs().max()) < 1e-6
            ok &= good
            print(f"{n} knots x {D} dof, {profile}, dt {dt}: {'ok' if good else 'DIFFERENT ' + str(float((ref[:, 0] - ours).abs().max()))}")
sys.exit(0 if ok else 1)
