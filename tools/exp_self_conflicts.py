"""experiment: self-collision pass time of the fused kernel vs LDS bank conflicts of the pair list order"""
import os, sys, dataclasses
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd._lib import load
from curobo_amd.robot import load_packaged_robot
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world, seed_knots, start_configuration

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin0 = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
B = 1024
lib = load()

def conflicts(pairs):
    tot = 0
    for k0 in range(0, len(pairs) - 15, 16):
        for c in (0, 1):
            _, cnt = np.unique(pairs[k0:k0 + 16, c] % 16, return_counts=True)
            tot += cnt.max()
    return tot / (2 * (len(pairs) // 16))

def run(pairs, label):
    sc = dataclasses.replace(kin0.self_collision, collision_pairs=torch.as_tensor(pairs.astype(np.int16), device=dev))
    kin = dataclasses.replace(kin0, self_collision=sc)
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg())
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=dev).reshape(B, -1)
    for _ in range(3):
        ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ro.cost_and_gradient(x)
    e1.record(); torch.cuda.synchronize()
    prof = torch.zeros(B, 16, dtype=torch.int64, device=dev)
    lib.curobo_hip_rollout_fused_set_profile_buffer(prof.data_ptr())
    ro.cost_and_gradient(x); torch.cuda.synchronize()
    lib.curobo_hip_rollout_fused_set_profile_buffer(None)
    t = prof.cpu().numpy().astype(np.float64) / 100.0
    d = t[:, 5] - t[:, 2]; d = d[np.abs(d) < 1e6]
    print(f"{label:28s} max-way conflict {conflicts(pairs):.2f}  launch {e0.elapsed_time(e1)*50:.1f} us  P2 self {d.mean():.2f} us  P2 total {(t[:,3]-t[:,2]).mean():.2f}")

orig = model.collision_pairs.astype(np.int64)
run(orig, "original order")
rng = np.random.default_rng(0)
run(orig[rng.permutation(len(orig))], "random order")
# greedy conflict-free blocks: each block of 16 pairs has distinct i%16 and distinct j%16 (endpoints may swap)
def schedule(pairs):
    left = [tuple(p) for p in pairs]
    out = []
    while left:
        ui, uj, blk, rest = set(), set(), [], []
        for (i, j) in left:
            if len(blk) < 16 and (i % 16) not in ui and (j % 16) not in uj:
                blk.append((i, j)); ui.add(i % 16); uj.add(j % 16)
            elif len(blk) < 16 and (j % 16) not in ui and (i % 16) not in uj:
                blk.append((j, i)); ui.add(j % 16); uj.add(i % 16)
            else:
                rest.append((i, j))
        while len(blk) < 16 and rest:  # fill with whatever is left (conflicts)
            blk.append(rest.pop())
        out += blk; left = rest
    return np.array(out)
sch = schedule(orig)
assert len(sch) == len(orig)
run(sch, "greedy conflict-free blocks")
