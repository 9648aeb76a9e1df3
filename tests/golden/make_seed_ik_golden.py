"""Golden vectors for the seed-IK iteration-state update, produced by the REFERENCE's own code
(curobo/_src/solver/seed_ik/seed_iteration_state_manager.py, pure torch, runs on CPU):
    PYTHONPATH=/root/reference python tests/golden/make_seed_ik_golden.py
Candidate / current states are random; trust ratios within 1e-3 (relative) of rho_min are
re-drawn so that the accept / reject decision does not depend on rounding."""
import os

import numpy as np
import torch
from curobo._src.solver.seed_ik.seed_ik_state import SeedIKState
from curobo._src.solver.seed_ik.seed_iteration_state_manager import SeedIterationStateManager

rng = np.random.default_rng(20)
n, D, T = 160, 7, 2
R = 6 * T + D
cfg = dict(rho_min=1e-3, lambda_factor=2.0, lambda_min=1e-5, lambda_max=1e10, convergence_position_tolerance=1e-5,
           convergence_orientation_tolerance=1e-5, convergence_joint_limit_weight=1.0)
lo, hi = -np.ones(D, np.float32), np.ones(D, np.float32)
f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
cur = dict(joint_position=0.7 * f(n, D), jacobian=f(n, R, D), jTerror=f(n, D), error_norm=np.abs(f(n)) + 0.5,
           position_errors=np.abs(f(n)) * 1e-5, orientation_errors=np.abs(f(n)) * 1e-5, lambda_damping=np.abs(f(n)) + 0.01)
cand = dict(joint_position=0.8 * f(n, D), jacobian=f(n, R, D), jTerror=f(n, D), error_norm=np.abs(f(n)) + 0.3,
            position_errors=np.abs(f(n)) * 1e-5, orientation_errors=np.abs(f(n)) * 1e-5)
pred = (f(n) * 0.5).astype(np.float32)
pred[:4] = 0.0
rho = (cur["error_norm"] - cand["error_norm"]) / (pred + np.float32(1e-8))
edge = np.abs(rho - cfg["rho_min"]) < 1e-3 * (1 + np.abs(rho))
cand["error_norm"][edge] += 0.05
mgr = SeedIterationStateManager(action_min=torch.tensor(lo), action_max=torch.tensor(hi), **cfg)
t = torch.as_tensor
cs = SeedIKState(joint_position=t(cur["joint_position"]), jacobian=t(cur["jacobian"]), jTerror=t(cur["jTerror"]),
                 error_norm=t(cur["error_norm"]), position_errors=t(cur["position_errors"]),
                 orientation_errors=t(cur["orientation_errors"]), lambda_damping=t(cur["lambda_damping"]).view(n, 1, 1))
ns = SeedIKState(joint_position=t(cand["joint_position"]), jacobian=t(cand["jacobian"]), jTerror=t(cand["jTerror"]),
                 error_norm=t(cand["error_norm"]), position_errors=t(cand["position_errors"]),
                 orientation_errors=t(cand["orientation_errors"]))
out = mgr.update_iteration_state(current_state=cs, candidate_state=ns, predicted_reduction=t(pred), batch_size=n)
res = dict(joint_position=out.joint_position, jacobian=out.jacobian, jTerror=out.jTerror, error_norm=out.error_norm,
           position_errors=out.position_errors, orientation_errors=out.orientation_errors,
           lambda_damping=out.lambda_damping.view(n), success=out.success, improvement=out.improvement)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seed_ik_update_golden.npz")
np.savez_compressed(path, lo=lo, hi=hi, pred=pred, **{f"cfg/{k}": np.float64(v) for k, v in cfg.items()},
                    **{f"cur/{k}": v for k, v in cur.items()}, **{f"cand/{k}": v for k, v in cand.items()},
                    **{f"out/{k}": np.asarray(v) for k, v in res.items()})
print(path, os.path.getsize(path), "accepted", int(res["improvement"].sum()), "of", n, "converged", int(res["success"].sum()))
