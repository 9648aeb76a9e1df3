import sys, time
sys.path.insert(0, "/root/repo")
import torch
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver import IKSolver, IKSolverCfg
from curobo_amd.workloads import c1_world, reachable_goals
dev = torch.device("cuda:0")
kin = KinematicsParams.from_model(load_packaged_robot("franka"), dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
P, S = 100, 64
gp, gq = reachable_goals(kin, P, seed=7)
for kw in (dict(), dict(use_lm_seed=True), dict(stream_shards=2), dict(stream_shards=4), dict(use_lm_seed=True, stream_shards=4)):
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, **kw))
    res = solver.solve_pose(gp, gq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        res = solver.solve_pose(gp, gq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(kw, f"{dt*1e3:.2f} ms/batch  {P/dt:.0f} solves/s  success {res.success.float().mean().item():.2f}  med pos err {res.position_error.median().item():.2e}")
