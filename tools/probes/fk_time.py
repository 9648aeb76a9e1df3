"""FK forward (spheres + cumulative transforms) and FK VJP at the C4 size (Unitree G1, 33 792 points; argv: robot, points): us per
launch (hipGraph replay) and checksums.  Knobs under test: CUROBO_FK_BWD_ROWS / CUROBO_FK_BWD_PTS."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B_  # noqa: E402
from curobo_amd.backends import kinematics as K  # noqa: E402
from curobo_amd.kinematics import KinematicsCfg  # noqa: E402

dev = torch.device("cuda:0")
robot = sys.argv[1] if len(sys.argv) > 1 else "unitree_g1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 33792
H = 33
k = KinematicsCfg.from_packaged(robot, device=dev).kinematics_config
L, D, S, T = k.num_links, k.num_dof, k.num_spheres, k.num_pose_links
g = torch.Generator().manual_seed(0)
lo, hi = k.joint_limits_position[0].cpu(), k.joint_limits_position[1].cpu()
q = (lo + (hi - lo) * torch.rand(n, D, generator=g)).to(dev)
z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
pos, quat, sph, com, cumul = z(n, T, 3), z(n, T, 4), z(n, S, 4), z(n, 4), z(n, L, 3, 4)
env = torch.zeros(n // H, dtype=torch.int32, device=dev)
fwd = lambda: K.launch_kinematics_forward_spheres(pos, quat, sph, com, cumul, q, k.fixed_transforms, k.link_spheres, k.link_masses_com,  # noqa: E731
                                                  k.joint_map_type, k.joint_map, k.link_map, k.tool_frame_map, k.link_sphere_idx_map,
                                                  k.joint_offset_map, env, k.num_envs, n, H, D, S, 32, True, False)
gq, gpos, gquat = z(n, D), torch.randn(n, T, 3, generator=g).to(dev), torch.randn(n, T, 4, generator=g).to(dev)
gs = z(n, S, 4)
idx = torch.randint(0, S, (n, 2), generator=g).to(dev)  # the self-collision gradient: two spheres per point
gs.scatter_(1, idx.view(n, 2, 1).expand(n, 2, 4), torch.randn(n, 2, 4, generator=g).to(dev))
bwd = lambda: K.launch_kinematics_backward(gq, gpos, gquat, gs, com, com, gpos, cumul, k.link_spheres, k.link_masses_com, k.link_map,  # noqa: E731
                                           k.joint_map, k.joint_map_type, k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data,
                                           k.link_chain_offsets, k.joint_links_data, k.joint_links_offsets, k.joint_affects_endeffector,
                                           k.joint_offset_map, env, k.num_envs, n, H, D, S, False, False)
fwd(); bwd(); torch.cuda.synchronize()
out = {}
for name, fn in (("forward", fwd), ("backward", bwd)):
    gr = B_.graphed(fn, 3, torch)
    out[name] = round(B_.time_kernel(gr.replay, 3, torch, min_s=0.05) / 3, 1)
print(os.environ.get("CUROBO_FK_BWD_ROWS", "-"), os.environ.get("CUROBO_FK_BWD_PTS", "-"), robot, n, out,
      "checksums", f"{float(sph.abs().sum()):.6e}", f"{float(gq.abs().sum()):.6e}")
