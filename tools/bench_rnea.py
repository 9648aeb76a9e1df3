"""RNEA forward / backward at the C4 size (Unitree G1, 33 792 elements): us per launch (hipGraph replay), and a check against
the values of the first variant run in the process.  Usage: python tools/bench_rnea.py [robot] [elements]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B_  # noqa: E402
from curobo_amd.backends import dynamics as Dy  # noqa: E402
from curobo_amd.kinematics import KinematicsCfg  # noqa: E402

dev = torch.device("cuda:0")
robot = sys.argv[1] if len(sys.argv) > 1 else "unitree_g1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 33792
kcfg = KinematicsCfg.from_packaged(robot, device=dev)
kin = kcfg.kinematics_config
L, D = kin.num_links, kin.num_dof
g = torch.Generator().manual_seed(0)
lo, hi = kin.joint_limits_position[0].cpu(), kin.joint_limits_position[1].cpu()
q = (lo + (hi - lo) * torch.rand(n, D, generator=g)).to(dev)
qd, qdd = torch.randn(n, D, generator=g).to(dev) * 0.5, torch.randn(n, D, generator=g).to(dev)
gt = torch.randn(n, D, generator=g).to(dev)
grav = torch.tensor([0, 0, 0, 0, 0, 9.81], device=dev)
tau, cache, ws = torch.zeros(n, D, device=dev), torch.zeros(n, L * 20, device=dev), torch.zeros(n, L * 18, device=dev)
gs = [torch.zeros(n, D, device=dev) for _ in range(3)]
rargs = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
         kin.joint_offset_map, grav, kin.link_level_offsets, kin.link_level_data)
fwd = lambda: Dy.launch_rnea_forward(tau, q, qd, qdd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None)  # noqa: E731
bwd = lambda: Dy.launch_rnea_backward(*gs, gt, q, qd, *rargs, cache, n, L, D, kin.n_tree_levels, 1, None, ws)  # noqa: E731
fwd(); bwd(); torch.cuda.synchronize()
out = {}
for name, fn in (("forward", fwd), ("backward", bwd)):
    gr = B_.graphed(fn, 3, torch)
    out[name] = round(B_.time_kernel(gr.replay, 3, torch, min_s=0.05) / 3, 1)
chk = [float(tau.abs().sum()), float(gs[0].abs().sum()), float(gs[1].abs().sum()), float(gs[2].abs().sum())]
print(os.environ.get("CUROBO_RNEA_LANES", "default"), robot, n, out, "checksums", [f"{c:.6e}" for c in chk])
