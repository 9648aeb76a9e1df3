"""C4 rollout set (Unitree G1, 1024 rollouts x 33 points, kernel sequence): the whole batch as one launch set against the
same rows as N sub-batches on N HIP streams inside one hipGraph (rows are independent: same results).   [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B_  # noqa: E402
from curobo_amd.kinematics import KinematicsCfg  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.workloads import seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
kcfg = KinematicsCfg.from_packaged("unitree_g1", device=dev)
model, kin = kcfg.model, kcfg.kinematics_config
B = 1024
x = torch.as_tensor(seed_knots(model, B, 12, seed=6, spread=0.15), device=dev).reshape(B, -1)
start = torch.as_tensor(start_configuration(model), device=dev)
kw = dict(use_fused=False, use_torque_limits=True, effort_limit=[200.0] * kin.num_dof)
ref = None
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    rows = B // n
    ros, streams, xs = [], [], []
    for i in range(n):
        # (a side stream inside every sub-batch's stream -- a two-level fork inside one capture -- crashes hipStreamEndCapture here:
        # the sub-batches run their joint-space chain on their own stream)
        ro = TrajOptRollout(kin, None, rows, TrajOptRolloutCfg(**kw, overlap_dynamics=(n == 1)))
        ro.update_start_state(start)
        ros.append(ro)
        streams.append(torch.cuda.Stream(device=dev))
        xs.append(x[i * rows:(i + 1) * rows].contiguous())
    outs = [None] * n

    def run():
        cur = torch.cuda.current_stream(dev)
        if n == 1:
            outs[0] = ros[0].cost_and_gradient(xs[0])
            return
        for i in range(n):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs[i] = ros[i].cost_and_gradient(xs[i])
        for i in range(n):
            cur.wait_stream(streams[i])
    run(); run()
    torch.cuda.synchronize()
    g = B_.graphed(run, 2, torch)
    us = B_.time_kernel(g.replay, 2, torch, min_s=0.1) / 2
    cost = torch.cat([o[0].reshape(-1) for o in outs]).clone()
    grad = torch.cat([o[1].reshape(rows, -1) for o in outs]).clone()
    if ref is None:
        ref = (cost, grad)
    print(f"{n} sub-batches of {rows} rollouts: {us:8.1f} us per rollout set;  max |cost - one batch| {float((cost - ref[0]).abs().max()):.3e}, "
          f"max |grad - one batch| {float((grad - ref[1]).abs().max()):.3e}", flush=True)
