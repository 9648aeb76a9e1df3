"""Self collision at scale against the oracle: many random configurations per robot (inside, at and beyond the joint limits),
the arg-max pair (flags) exact, distance and gradient to rounding.   python tests/randomised/fuzz_self.py [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402

from curobo_amd.backends import geometry as G  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bad = 0
for robot, n in (("franka", 20000), ("ur10e", 20000), ("unitree_g1", 2500)):
    model = load_model(robot)
    kp = KinematicsParams.from_model(model, dev)
    S, P = model.num_spheres, model.collision_pairs.shape[0]
    for scale in (0.4, 1.0, 1.4):
        t0 = time.time()
        q = sample_q(model, n, seed=seed * 17 + int(scale * 10)) * scale
        sph = oracle.kinematics_forward(q, model.as_dict())["robot_spheres"]
        ref = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 2.5)
        out_d = torch.full((n, 1), -1.0, device=dev)
        out_g = torch.zeros(n, S, 4, device=dev)
        flags = torch.zeros(n, S, dtype=torch.uint8, device=dev)
        G.self_collision_distance(out_d, out_g, torch.zeros(1, device=dev), flags, torch.as_tensor(sph, device=dev),
                                  kp.self_collision.sphere_padding, torch.tensor([2.5], device=dev), kp.self_collision.collision_pairs,
                                  torch.zeros(1, device=dev), torch.zeros(2, dtype=torch.int16, device=dev), 1, 256, n, 1, S, P, False, True)
        torch.cuda.synchronize()
        d, g, f = out_d.cpu().numpy()[:, 0], out_g.cpu().numpy(), flags.cpu().numpy()
        n_col = int((ref["distance"] > 0).sum())
        flags_off = int((f != ref["sparse_index"]).any(-1).sum())
        d_err = float(np.abs(d - ref["distance"]).max())
        g_err = float(np.abs(g - ref["gradient"]).max())
        # a flag mismatch is a tie broken differently only if the two pairs' penetrations agree to rounding: report the worst
        status = "ok" if flags_off == 0 and d_err < 2e-5 * max(1.0, float(np.abs(ref["distance"]).max())) else "MISMATCH"
        bad += status != "ok"
        print(f"{robot:11s} scale {scale}: {n} configurations, {n_col} in self collision; flag rows off {flags_off}, max |d - ref| {d_err:.2e}, "
              f"max |g - ref| {g_err:.2e}  [{status}]  ({time.time() - t0:.1f} s)", flush=True)
print("mismatching sets:", bad)
