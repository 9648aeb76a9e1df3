"""a large mesh (a bumpy ball of ~160 k triangles): build time and memory of the BVH + cell lists, the launch on robot trajectories
through it, and the distances against the oracle's brute force on a sample"""
import os
import sys
import time

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
from conftest import load_model  # noqa: E402
from test_oracle_mesh import sphere_shape  # noqa: E402

import curobo_amd.backends.mesh as MB  # noqa: E402
from curobo_amd.scene import MeshStore  # noqa: E402
from oracle.oracle import Oracle, mesh_scene_arrays  # noqa: E402

dev = torch.device("cuda:0")
orc = Oracle()
nu, nv = int(os.environ.get("NU", "200")), int(os.environ.get("NV", "400"))
v, f = sphere_shape(0.35, nu, nv)
r = np.linalg.norm(v, axis=1, keepdims=True)
d = v / r
v = (v * (1.0 + 0.08 * np.sin(9 * d[:, :1]) * np.sin(7 * d[:, 1:2]) * np.sin(5 * d[:, 2:3]))).astype(np.float32)  # bumps: no longer equidistant
world = [[{"name": "rock", "vertices": v, "faces": f, "pose": [0.45, 0.1, 0.35, 1, 0, 0, 0]}]]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
store = MeshStore(world, dev, cells={"0": False, "1": None}[os.environ.get("CELLS", "1")])
torch.cuda.synchronize()
print(f"triangles {len(f)}  build {time.perf_counter() - t0:.2f} s  device memory {(free0 - torch.cuda.mem_get_info()[0]) / 2**20:.0f} MiB  cells {getattr(store.meshes[0], 'cells_info', None)}", flush=True)
model = load_model("franka")
rng = np.random.default_rng(3)
B, H = 128, 33
lo, hi = model.joint_limits_position
q0, q1 = rng.uniform(lo, hi, size=(B, 1, 7)) * 0.6, rng.uniform(lo, hi, size=(B, 1, 7)) * 0.6
tt = np.linspace(0, 1, H)[None, :, None]
q = (q0 * (1 - tt) + q1 * tt).astype(np.float32)
sph = orc.kinematics_forward(q.reshape(B * H, 7), model.as_dict(), horizon=H)["robot_spheres"].reshape(B, H, -1, 4)
S = sph.shape[2]
t = torch.as_tensor(sph, device=dev)
w, eta, dt = torch.tensor([1.0], device=dev), torch.tensor([0.02], device=dev), torch.tensor([0.05], device=dev)
dist, grad = torch.zeros(B, H, S, device=dev), torch.zeros(B, H, S, 4, device=dev)
for _ in range(2):
    MB.sphere_mesh_collision(dist, grad, t, store.struct, w, eta, None, B, H, S, False, 3, True, dt, accumulate=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    MB.sphere_mesh_collision(dist, grad, t, store.struct, w, eta, None, B, H, S, False, 3, True, dt, accumulate=False)
e1.record()
torch.cuda.synchronize()
cnt = next(iter(dist._curobo_mesh_ws.values()))[:16].view(torch.int32).tolist()
print(f"launch {e0.elapsed_time(e1) * 100:.1f} us  counters [heavy, to walk, light, to wide] {cnt}  spheres in collision {int((dist > 0).sum())} of {dist.numel()}", flush=True)
# parity on a sample of trajectories
sel = np.arange(0, B, 16)
t0 = time.perf_counter()
ref = orc.scene_collision(sph[sel], mesh_scene_arrays(world), 1.0, 0.02, sweep=True, enable_speed_metric=True, speed_dt=0.05)
got = dist[sel].cpu().numpy()
err = np.abs(got - ref["distance"])
print(f"oracle on {len(sel)} trajectories {time.perf_counter() - t0:.1f} s   max |diff| {err.max():.2e} (cost up to {ref['distance'].max():.2f})  hit sets differ on {int(((got > 0) != (ref['distance'] > 0)).sum())}", flush=True)
