"""development: where the time of one collision-free IK batch (C1, exit_early) goes"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.solver import IKSolver, IKSolverCfg
from curobo_amd.workloads import c1_world, feasible_goals

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
P, S = 100, 64
solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, stream_shards=4))
gp, gq = feasible_goals(kin, scene, P)
def sync(): torch.cuda.synchronize()
def timeit(fn, n=20):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3
print("solve_pose (exit_early)        %.3f ms" % timeit(lambda: solver.solve_pose(gp, gq, exit_early=True)))
T, G, D = kin.num_pose_links, 1, kin.num_dof
gpe = gp.to(dev).view(P, 1, G, 3).expand(P, T, G, 3).contiguous(); gqe = gq.to(dev).view(P, 1, G, 4).expand(P, T, G, 4).contiguous()
ss = solver.seed_solver
print("seed_solver.solve_batch         %.3f ms" % timeit(lambda: ss.solve_batch(gpe, gqe, return_seeds=S)))
def inner_only():
    ss._run_inner()
print("  one _run_inner (4 LM iterations, graph) %.3f ms" % timeit(inner_only))
seeds = ss.solve_batch(gpe, gqe, return_seeds=S).solution
print("  iterations used:", ss.solve_batch(gpe, gqe, return_seeds=S).iterations)
print("_get_result (metrics + ranking)  %.3f ms" % timeit(lambda: solver._get_result(seeds.reshape(P * S, D).contiguous(), 1)))
def goals():
    for ro, rows in zip(solver.rollouts, solver._row_goals): ro.update_goals(gpe, gqe, rows)
    solver.metrics_rollout.update_goals(gpe, gqe, solver._mrow_goal)
print("goal updates                     %.3f ms" % timeit(goals))
print("generate_seeds                   %.3f ms" % timeit(lambda: ss.generate_seeds(None)))
