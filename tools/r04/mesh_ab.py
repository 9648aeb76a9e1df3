"""One mesh collision launch on the bench's mesh world (the C2 world's cuboids as triangle meshes), timed, outputs saved.

    CUROBO_MESH_WALK=1|2 python tools/r04/mesh_ab.py <out.npz>          (1 = the eight-at-a-time walk, 2 = the self-scheduled walk)
    python tools/r04/mesh_ab.py --compare a.npz b.npz                    (bitwise)
"""
import os
import sys
import time

import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        ok &= same
        print(k, "identical" if same else f"DIFFERENT: max abs {np.abs(a[k] - b[k]).max():.3e}, {int((a[k] != b[k]).sum())} of {a[k].size} values")
    sys.exit(0 if ok else 1)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
B, H = int(os.environ.get("BATCH", "1024")), 33
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


out = {}
for sweep in (3, 0):
    for _ in range(3):
        scene_pass(sweep)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        scene_pass(sweep)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20))
    out[f"dist_sweep{sweep}"] = ro.scene_dist.detach().cpu().numpy().copy()
    out[f"grad_sweep{sweep}"] = ro.scene_grad.detach().cpu().numpy().copy()
    print(f"walk mode {os.environ.get('CUROBO_MESH_WALK', 'default')} sweep {sweep}: median {ts[10]:.1f} us, min {ts[0]:.1f} us per launch (select + walk);"
          f" cost sum {float(ro.scene_dist.sum()):.6e}", flush=True)
    if hasattr(lib, "curobo_hip_mesh_stats"):
        import ctypes as C
        buf = (C.c_ulonglong * 8)()
        lib.curobo_hip_mesh_stats(buf, 1)
        scene_pass(sweep)
        torch.cuda.synchronize()
        lib.curobo_hip_mesh_stats(buf, 1)
        names = ["closest calls", "steps", "leaf visits", "wave passes", "transitions", "full queries", "items", "item-slots"]
        print("   ", {n: int(v) for n, v in zip(names, buf)}, flush=True)
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **out)
