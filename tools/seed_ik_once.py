"""One warm seed-IK solve (100 problems x 128 LM seeds, Franka) -- target for rocprofv3 runs
(kernel trace of the five-launch LM iteration; MFMA counters of lm_step_kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.solver.seed_ik import SeedIKSolver, SeedIKSolverCfg  # noqa: E402
from curobo_amd.workloads import reachable_goals  # noqa: E402

dev = torch.device("cuda:0")
kin = KinematicsParams.from_model(load_packaged_robot("franka"), dev)
P, S = 100, 128
solver = SeedIKSolver(kin, P, SeedIKSolverCfg(num_seeds=S, batch_success_threshold=2.0))
gp, gq = reachable_goals(kin, P, seed=7)
res = solver.solve_batch(gp.view(P, 1, 3), gq.view(P, 1, 4))
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    res = solver.solve_batch(gp.view(P, 1, 3), gq.view(P, 1, 4))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"seed IK: {P} problems x {S} seeds, 16 LM iterations: {dt * 1e3:.2f} ms per batch, "
      f"{P * S * 16 / dt / 1e6:.2f} M LM iterations/s, success {res.success[:, 0].float().mean().item():.2f}")
