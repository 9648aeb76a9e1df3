"""Where a TrajOptSolver.solve_pose call (1 problem x 8 seeds, C2 world) spends its time: host-synchronised phase timers"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
from curobo_amd.workloads import c2_world, feasible_goals, start_configuration
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
start = torch.as_tensor(start_configuration(model))
P, S = int(os.environ.get("P", "1")), int(os.environ.get("S", "8"))
slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
gp, gq = feasible_goals(kin, scene, 64)
gp, gq = gp[:P].contiguous(), gq[:P].contiguous()
for _ in range(3):
    r = slv.solve_pose(start, gp, gq)
torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    return w
slv.ik.solve_pose = timed("ik.solve_pose", slv.ik.solve_pose)
slv.optimizer.optimize = timed("optimizer.optimize", slv.optimizer.optimize)
slv._seed_metrics = timed("_seed_metrics", slv._seed_metrics)
slv._rank = timed("_rank", slv._rank)
slv._set_problem = timed("_set_problem", slv._set_problem)
slv.seed_knots = timed("seed_knots", slv.seed_knots)
slv.compute_trajectory_dt = timed("compute_trajectory_dt", slv.compute_trajectory_dt)
reps = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    r = slv.solve_pose(start, gp, gq)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / reps
print(f"P={P} S={S}: {tot * 1e3:.2f} ms per solve (with the phase synchronisations), success {float(r.success.float().mean()):.2f}, passes {r.finetune_passes}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / reps * 1e3:7.3f} ms")
print(f"  {'(rest)':28s} {(tot - sum(acc.values()) / reps) * 1e3:7.3f} ms")
