import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver import IKSolver, IKSolverCfg
from curobo_amd.workloads import c1_world, feasible_goals
dev = torch.device("cuda:0")
model = load_packaged_robot("franka"); kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
solver = IKSolver(kin, scene, 100, IKSolverCfg(num_seeds=64, stream_shards=4))
gp, gq = feasible_goals(kin, scene, 100)
for _ in range(60): r = solver.solve_pose(gp, gq, exit_early=True)
torch.cuda.synchronize(); print(float(r.success.float().mean()))
