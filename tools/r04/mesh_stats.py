"""Work counters of one mesh collision launch on the bench's mesh world (needs a -DCUROBO_MESH_STATS build of the library)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from curobo_amd._lib import load  # noqa: E402
from curobo_amd.backends import collision as Cn  # noqa: E402
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, box_mesh  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)


def subdivide(v, f, times):
    v = [tuple(x) for x in np.asarray(v, np.float64)]
    f = np.asarray(f, np.int64)
    for _ in range(times):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                cache[key] = len(v)
                v.append(tuple((np.asarray(v[a]) + np.asarray(v[b])) * 0.5))
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
        f = np.asarray(out, np.int64)
    return np.asarray(v, np.float32), f.astype(np.int32)


world = c2_world()
sub = int(os.environ.get("SUBDIV", "4"))
meshes = [[dict(name=f"box{i}", pose=o["pose"], **dict(zip(("vertices", "faces"), subdivide(*box_mesh(o["dims"]), sub))))
           for i, o in enumerate(world[0])]]
B, H = 1024, 33
cfg = CollisionRolloutCfg(use_fused=False)
x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
scene = SceneData.from_arrays(None, dev, meshes=meshes)
ro = CollisionRollout(kin, scene, B, cfg)
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
ro.compute_kinematics(ro.compute_state_from_action(x.view(B, cfg.n_knots, -1)))
S = kin.num_spheres
lib = load()


def scene_pass(sweep=3):
    Cn.sphere_obstacle_collision(ro.scene_dist, ro.scene_grad, ro.robot_spheres, scene.struct, ro._w_scene, ro._eta, ro.env_query_idx,
                                 B, cfg.padded_horizon, S, False, sweep, sweep > 0, ro._speed_dt)


for sweep in (3, 0):
    scene_pass(sweep)
    torch.cuda.synchronize()
    if hasattr(lib, "curobo_hip_mesh_stats"):
        buf = (C.c_ulonglong * 8)()
        lib.curobo_hip_mesh_stats(buf, 1)
        lanes = np.zeros(1 << 18, np.uint32)
        lib.curobo_hip_mesh_lane_stats(None, 1)
        scene_pass(sweep)
        torch.cuda.synchronize()
        lib.curobo_hip_mesh_stats(buf, 1)
        lib.curobo_hip_mesh_lane_stats(lanes.ctypes.data_as(C.c_void_p), 1)
        moves, trans = lanes[:1 << 17].astype(np.int64), lanes[1 << 17:].astype(np.int64)
        live = moves + trans
        live = live[live > 0]
        per_wave = (moves + trans)[: (live.size + 7) // 8 * 8].reshape(-1, 8).max(1)
        print(f"  per item passes (moves + transitions): n {live.size} mean {live.mean():.1f} median {np.median(live):.0f} p90 {np.percentile(live, 90):.0f} "
              f"p99 {np.percentile(live, 99):.0f} p99.9 {np.percentile(live, 99.9):.0f} max {live.max()};  per wavefront (max of 8 consecutive items): mean {per_wave.mean():.1f} "
              f"p90 {np.percentile(per_wave, 90):.0f} p99 {np.percentile(per_wave, 99):.0f} max {per_wave.max()}")
        top = np.argsort(moves + trans)[::-1][:5]
        print("  heaviest items: passes", (moves + trans)[top], "queries+1", trans[top])
        names = ["closest calls", "steps", "leaf visits", "wave passes", "transitions", "full queries", "items", "item-slots"]
        print(f"sweep {sweep}: spheres {B * H * S}", {n: int(v) for n, v in zip(names, buf)})
    t0 = time.perf_counter()
    for _ in range(5):
        scene_pass(sweep)
    torch.cuda.synchronize()
    print(f"sweep {sweep}: {(time.perf_counter() - t0) / 5 * 1e6:.1f} us per launch (host timed), cost sum {float(ro.scene_dist.sum()):.6e}")
