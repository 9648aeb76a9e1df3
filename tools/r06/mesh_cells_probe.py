"""The queued mesh launch on the bench's mesh world (the C2 world's cuboids as triangle meshes, 1024 x 33 points of Franka spheres,
swept + speed metric), with and without the cell lists: times, the queue counters (how many live spheres the cell lists handed
to the tree walk), and bitwise comparison of the outputs.  Run it under `rocprofv3 --kernel-trace --stats` for per-kernel times.

    python tools/r06/mesh_cells_probe.py [reps]         (env: CELL_SIZE, PAD, CAP, BATCH)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/")
from bench import c2_world_as_meshes  # noqa: E402
from curobo_amd.backends import mesh as M  # noqa: E402
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import MeshStore, SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
B = int(os.environ.get("BATCH", "1024"))
cfg = CollisionRolloutCfg(use_fused=False)
ro = CollisionRollout(kin, SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev), B, cfg)
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
ro.compute_kinematics(ro.compute_state_from_action(x.view(B, cfg.n_knots, -1)))
sph = ro.robot_spheres
H, S = cfg.padded_horizon, kin.num_spheres
meshes = c2_world_as_meshes()
kw = {"cell_size": float(os.environ.get("CELL_SIZE", "0.02")), "pad": float(os.environ.get("PAD", "0.2")), "gather_cap": int(os.environ.get("CAP", "2048"))}
w, eta, dt = ro._w_scene, ro._eta, ro._speed_dt
res = {}
for name, cells in (("tree_walk", False), ("cell_lists", kw)):
    store = MeshStore(meshes, dev, cells=cells)
    dist, grad = torch.zeros(B, H, S, device=dev), torch.zeros(B, H, S, 4, device=dev)

    def launch():
        M.sphere_mesh_collision(dist, grad, sph, store.struct, w, eta, None, B, H, S, False, 3, True, dt, accumulate=False)
    launch()
    torch.cuda.synchronize()
    ws = next(iter(dist._curobo_mesh_ws.values()))
    cnt = ws[:16].view(torch.int32).cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    res[name] = (dist.cpu().numpy(), grad.cpu().numpy())
    print(f"{name}: {us:.1f} us per launch; queue: heavy {cnt[0]}, light {cnt[2]}, handed to the tree walk {cnt[1]}; cost sum {float(dist.sum()):.6e}")
    if cells and hasattr(M.load(), "curobo_hip_mesh_stats"):
        import ctypes as C
        lib = M.load()
        buf = (C.c_ulonglong * 8)()
        lib.curobo_hip_mesh_stats.argtypes = [C.c_void_p, C.c_int]
        lib.curobo_hip_mesh_stats(C.cast(buf, C.c_void_p), 1)
        lib.curobo_hip_mesh_lane_stats.argtypes = [C.c_void_p, C.c_int]
        lib.curobo_hip_mesh_lane_stats(None, 1)
        launch()
        torch.cuda.synchronize()
        lib.curobo_hip_mesh_stats(C.cast(buf, C.c_void_p), 1)
        lane = (C.c_uint * (1 << 18))()
        lib.curobo_hip_mesh_lane_stats.argtypes = [C.c_void_p, C.c_int]
        lib.curobo_hip_mesh_lane_stats(C.cast(lane, C.c_void_p), 0)
        la = np.frombuffer(lane, np.uint32)
        rounds, queries = la[:1 << 17], la[1 << 17:]
        nq = int(cnt[0]) + int(cnt[2])
        print("    per queue entry (first 131072): rounds max %d p99 %d mean %.2f; queries max %d p99 %d mean %.2f; heavy-class mean rounds %.2f, light %.2f"
              % (rounds.max(), np.percentile(rounds[:min(nq, 1 << 17)], 99), rounds[:min(nq, 1 << 17)].mean(), queries.max(),
                 np.percentile(queries[:min(nq, 1 << 17)], 99), queries[:min(nq, 1 << 17)].mean(), rounds[:cnt[0]].mean(), rounds[cnt[0]:min(nq, 1 << 17)].mean()))
        print("    stats of one launch: queries %d, rounds %d, triangle tests %d, ray-sign calls %d, skipped by the cell bound %d, entries scanned %d" % tuple(buf[:6]))
    if cells:
        for m in store.meshes:
            print("   ", m.cells_info)
a, b = res["tree_walk"], res["cell_lists"]
print("distance: identical" if np.array_equal(a[0], b[0]) else f"distance: {int((a[0] != b[0]).sum())} of {a[0].size} differ, max abs {np.abs(a[0] - b[0]).max():.3e} (max {a[0].max():.3e})")
print("gradient: identical" if np.array_equal(a[1], b[1]) else f"gradient: {int((a[1] != b[1]).any(-1).sum())} of {a[1].shape[0] * a[1].shape[1] * a[1].shape[2]} spheres differ, max abs {np.abs(a[1] - b[1]).max():.3e}")
