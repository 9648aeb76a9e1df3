"""which spheres the cell lists hand to the tree walk in the tests' mesh world, per mesh, and what the walk then costs"""
import os
import sys

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
from conftest import load_model  # noqa: E402
from test_oracle_mesh import mesh_world  # noqa: E402

import curobo_amd.backends.mesh as MB  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.scene import MeshStore, SceneData  # noqa: E402
from curobo_amd.workloads import seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_model("franka")
kin = KinematicsParams.from_model(model, dev)
B = int(os.environ.get("B", "128"))
world = mesh_world()
x = torch.as_tensor(seed_knots(model, B, 12, seed=11), device=dev).reshape(B, -1)
ro = TrajOptRollout(kin, SceneData.from_arrays(None, dev, meshes=MeshStore(world, dev)), B, TrajOptRolloutCfg(fused_mesh_max_batch=0))
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
ro.cost_and_gradient(x)
torch.cuda.synchronize()
sph = ro.robot_spheres.clone()
H, S = sph.shape[1], sph.shape[2]
w, eta, dt = torch.tensor([1.0], device=dev), torch.tensor([0.02], device=dev), torch.tensor([0.05], device=dev)
for sel in [None] + list(range(len(world[0]))):
    ws = [[m for i, m in enumerate(world[0]) if sel is None or i == sel]]
    if not ws[0][0].get("enable", True):
        continue
    store = MeshStore(ws, dev)
    info = [(m.n_tri, getattr(m, "cells_info", None)) for m in store.meshes]
    dist, grad = torch.zeros(B, H, S, device=dev), torch.zeros(B, H, S, 4, device=dev)
    for _ in range(2):
        MB.sphere_mesh_collision(dist, grad, sph, store.struct, w, eta, None, B, H, S, False, 3, True, dt, accumulate=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        MB.sphere_mesh_collision(dist, grad, sph, store.struct, w, eta, None, B, H, S, False, 3, True, dt, accumulate=False)
    e1.record()
    torch.cuda.synchronize()
    cnt = next(iter(dist._curobo_mesh_ws.values()))[:16].view(torch.int32).tolist()
    print("meshes", "all" if sel is None else world[0][sel]["name"], "launch", round(e0.elapsed_time(e1) * 100, 1), "us  counters [heavy, to walk, light, to wide]", cnt[:4],
          "in collision", int((dist > 0).sum()), flush=True)
    for n, ci in info:
        print("     triangles", n, "cells", ci, flush=True)
