"""development: fused seed-IK iterations vs the five-launch sequence on the C1 batch"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver.seed_ik import SeedIKSolver, SeedIKSolverCfg
from curobo_amd.workloads import c1_world, feasible_goals

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
P, S = 100, 128
gp, gq = feasible_goals(kin, scene, P)
T = kin.num_pose_links
gpe = gp.to(dev).view(P, 1, 1, 3).expand(P, T, 1, 3).contiguous(); gqe = gq.to(dev).view(P, 1, 1, 4).expand(P, T, 1, 4).contiguous()
res = {}
for fused in (False, True):
    ss = SeedIKSolver(kin, P, SeedIKSolverCfg(num_seeds=S, fused_iterations=fused))
    r = ss.solve_batch(gpe, gqe, return_seeds=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ss.sampler.reset() if hasattr(ss.sampler, "reset") else None
        r2 = ss.solve_batch(gpe, gqe, return_seeds=8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    res[fused] = (r, ss.q.clone(), ss.success.clone(), ss.position_error.clone())
    print(f"fused={fused}: {ms:.3f} ms per solve_batch; success (best seed) {r.success[:, 0].float().mean().item():.3f}; "
          f"runs converged {ss.success.float().mean().item():.3f}; iterations {r.iterations}")
qa, qb = res[False][1], res[True][1]
d = (qa - qb).abs().max(dim=-1).values
print("per-run |q_fused - q_sequence|_inf: median %.2e, 90%% %.2e, max %.2e; runs equal to 1e-4: %.3f" % (
    d.median().item(), d.quantile(0.9).item(), d.max().item(), (d < 1e-4).float().mean().item()))
print("success flags equal:", (res[False][2] == res[True][2]).float().mean().item())
ss = SeedIKSolver(kin, P, SeedIKSolverCfg(num_seeds=S, fused_iterations=True))
ss.solve_batch(gpe, gqe, return_seeds=8)
torch.cuda.synchronize()
for iters in (1, 4, 16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ss._iterate_fused(iters); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): ss._iterate_fused(iters)
    e1.record(); torch.cuda.synchronize()
    print(f"seed_ik_iterate({iters} iterations): {e0.elapsed_time(e1) * 100:.1f} us per launch")
