"""Where a pose-to-pose TrajOptSolver.solve_pose goes (1 problem x 8 seeds): host wall time of its stages (development probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curobo_amd.kinematics import KinematicsCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
from curobo_amd.workloads import c2_world, feasible_goals, start_configuration
dev = torch.device("cuda:0")
kcfg = KinematicsCfg.from_packaged("franka", device=dev)
model, kin = kcfg.model, kcfg.kinematics_config
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
P, S = 1, 8
slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
gp, gq = feasible_goals(kin, scene, 64)
gp, gq = gp[:P].contiguous(), gq[:P].contiguous()
start = torch.as_tensor(start_configuration(model))
for _ in range(2):
    r = slv.solve_pose(start, gp, gq)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    r = slv.solve_pose(start, gp, gq)
torch.cuda.synchronize()
pr.disable()
print("ms per solve", (time.perf_counter() - t0) / 5 * 1e3, "passes", r.finetune_passes)
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
