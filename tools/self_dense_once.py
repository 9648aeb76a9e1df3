"""development / profiling target: the dense-bitmap self-collision kernel at C4 size (Unitree G1; B x 33 points, B = argv[1], default 256)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.backends import geometry as G
from curobo_amd.kinematics import Kinematics, KinematicsCfg
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams

dev = torch.device("cuda:0")
model = load_packaged_robot("unitree_g1")
kp = KinematicsParams.from_model(model, dev)
B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 33
n = B * H
S, P = model.num_spheres, int(kp.self_collision.collision_pairs.shape[0])
rng = np.random.default_rng(0)
lo, hi = np.asarray(model.joint_limits_position, np.float32)
q = torch.as_tensor((0.5 * (lo + hi) + 0.3 * (hi - lo) * (rng.random((n, lo.shape[0])) - 0.5)).astype(np.float32), device=dev)
kin = Kinematics(KinematicsCfg(kp, model))
sph = kin.compute_kinematics(q).robot_spheres.reshape(n, S, 4).contiguous()
sc = kp.self_collision
w = torch.tensor([1.0], device=dev)
out_d, out_g, flags = torch.zeros(n, 1, device=dev), torch.zeros(n, S, 4, device=dev), torch.zeros(n, S, dtype=torch.uint8, device=dev)
z1, z2 = torch.zeros(1, device=dev), torch.zeros(2, dtype=torch.int16, device=dev)
def run():
    G.self_collision_distance(out_d, out_g, z1, flags, sph, sc.sphere_padding, w, sc.collision_pairs, z1, z2, 1, 256, B, H, S, P, False, True)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f"self collision, {n} points x {P} pairs: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us; in collision {(out_d > 0).float().mean().item():.3f}")
