"""Host-side profile of one pose-to-pose solve (cProfile, after warm-up): where the time between the graph replays goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg  # noqa: E402
from curobo_amd.workloads import c2_world, feasible_goals, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
P, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 8)
slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
gp, gq = feasible_goals(kin, scene, 64)
gp, gq = gp[:P].contiguous(), gq[:P].contiguous()
start = torch.as_tensor(start_configuration(model))
for _ in range(4):
    slv.solve_pose(start, gp, gq)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    slv.solve_pose(start, gp, gq)
torch.cuda.synchronize()
print(f"solve {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    slv.solve_pose(start, gp, gq)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
