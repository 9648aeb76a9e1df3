"""When the eight wavefronts of a fused-rollout workgroup leave the collision pass (diagnostic; needs a library built with
CUROBO_HIP_EXTRA_FLAGS=-DCUROBO_FUSED_WAVE_STAMPS): the spread is what the barrier behind the pass waits for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curobo_amd._lib import load
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world, seed_knots, start_configuration
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ro = CollisionRollout(kin, scene, B, cfg)
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
for _ in range(3):
    ro.cost_and_gradient(x)
torch.cuda.synchronize()
prof = torch.zeros(B, 16, dtype=torch.int64, device=dev)
lib = load()
lib.curobo_hip_rollout_fused_set_profile_buffer(prof.data_ptr())
ro.cost_and_gradient(x)
torch.cuda.synchronize()
lib.curobo_hip_rollout_fused_set_profile_buffer(None)
t = prof.cpu().numpy().astype(np.float64) / 100.0
w = t[:, 8:16] - t[:, 2:3]          # wave finish relative to the start of P2
end = t[:, 3] - t[:, 2]             # P2 end (after gather) for reference
print(f"waves leave the main round after (us): mean {w.mean():.2f}  fastest {w.min(1).mean():.2f}  slowest {w.max(1).mean():.2f}  "
      f"slowest - mean {(w.max(1) - w.mean(1)).mean():.2f}; P2 total {end.mean():.2f}")
print("per wave index mean:", np.round(w.mean(0), 2))
