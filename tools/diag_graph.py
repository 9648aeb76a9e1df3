import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world, seed_knots, start_configuration
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
dev = torch.device("cuda:0")
model = load_packaged_robot("franka"); kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg()
for G in (1, 10):
    ro = CollisionRollout(kin, scene, 1024, cfg); ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    opt = LBFGSOpt(LBFGSOptCfg(num_problems=256, inner_iters=G), ro.cost_and_gradient, 12, 7, (kin.joint_limits_position[0], kin.joint_limits_position[1]), dev)
    opt.reinitialize(torch.as_tensor(seed_knots(model, 256, 12), device=dev))
    def ev(fn, n):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
        return e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n
    print("G", G, "eager step (event us, wall us)", ev(opt._opt_step, 100))
    opt.capture()
    r = ev(opt._graph.replay, 50)
    print("G", G, "graph replay per step (event us, wall us)", r[0] / G, r[1] / G)
