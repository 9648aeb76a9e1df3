"""One warm fused-rollout launch sequence on the C2 workload (target for rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg(use_fused="--no-fused" not in sys.argv)
B = 1024
ro = CollisionRollout(kin, scene, B, cfg)
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
for _ in range(10):
    ro.cost_and_gradient(x)
torch.cuda.synchronize()
