"""How long the metrics + ranking stage of an IK batch takes (graph replay), and a fused IK launch over the same rows."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.solver import IKSolver, IKSolverCfg  # noqa: E402
from curobo_amd.workloads import c1_world, feasible_goals  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
solver = IKSolver(kin, scene, 100, IKSolverCfg(num_seeds=64, stream_shards=4))
gp, gq = feasible_goals(kin, scene, 100)
for _ in range(3):
    r = solver.solve_pose(gp, gq, exit_early=True)
torch.cuda.synchronize()
q = torch.rand(100 * 64, kin.num_dof, device=dev)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("metrics + ranking (graph replay + result copies): %.1f us" % timeit(lambda: solver._get_result(q, 1)))
m = solver.metrics_rollout
g = torch.cuda.CUDAGraph()
m.cost_and_gradient_fused(q, with_metrics=True)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    m.cost_and_gradient_fused(q, with_metrics=True)
print("fused IK launch with metric outputs, 6400 rows (graph replay): %.1f us" % timeit(g.replay))
g2 = torch.cuda.CUDAGraph()
m.evaluate(q.view(6400, 1, -1), with_gradient=False)
torch.cuda.synchronize()
with torch.cuda.graph(g2):
    m.evaluate(q.view(6400, 1, -1), with_gradient=False)
print("kernel sequence of the metrics rollout (graph replay): %.1f us" % timeit(g2.replay))
print("whole solve: %.1f us" % timeit(lambda: solver.solve_pose(gp, gq, exit_early=True), 50))
