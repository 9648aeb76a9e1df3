import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench as B_
from curobo_amd.backends import dynamics as Dy
from curobo_amd.kinematics import KinematicsCfg
dev = torch.device("cuda:0")
kin = KinematicsCfg.from_packaged("unitree_g1", device=dev).kinematics_config
n = 33792; L, D = kin.num_links, kin.num_dof
g = torch.Generator().manual_seed(0)
lo, hi = kin.joint_limits_position[0].cpu(), kin.joint_limits_position[1].cpu()
q = (lo + (hi - lo) * torch.rand(n, D, generator=g)).to(dev)
qd, qdd, gt = [torch.randn(n, D, generator=g).to(dev) for _ in range(3)]
grav = torch.tensor([0, 0, 0, 0, 0, 9.81], device=dev)
tau, cache, ws = torch.zeros(n, D, device=dev), torch.zeros(n, L * 20, device=dev), torch.zeros(n, L * 18, device=dev)
gs = [torch.zeros(n, D, device=dev) for _ in range(3)]
sc = torch.zeros(3 * n * D, device=dev)
rargs = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map, kin.joint_offset_map, grav, kin.link_level_offsets, kin.link_level_data)
for name, s in (("staged", None), ("scratch", sc)):
    fwd = lambda: Dy.launch_rnea_forward(tau, q, qd, qdd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None, scratch=s)
    bwd = lambda: Dy.launch_rnea_backward(*gs, gt, q, qd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None, ws, scratch=s)
    fwd(); bwd(); torch.cuda.synchronize()
    o = {}
    for nm, fn in (("forward", fwd), ("backward", bwd)):
        gr = B_.graphed(fn, 3, torch)
        o[nm] = round(B_.time_kernel(gr.replay, 3, torch, min_s=0.05) / 3, 1)
    print(name, o, float(tau.abs().sum()), float(gs[0].abs().sum()))
