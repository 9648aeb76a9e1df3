"""The ORACLE's tool-pose (goal-set) cost against the reference's own Warp kernel (wp_tool_pose.py, run through
tests/golden/warp_emulator) on RANDOM inputs: goal-set sizes 1-5, 1-3 links, horizons 1-6, all three rotation methods, axes
weights with zeros, tolerances, projection into the goal frame, poses at / near / far from their goals on both quaternion
hemispheres, shared goals.  CPU only, needs /root/reference.
    python tests/randomised/sweep_reference_warp_tool_pose.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/cost"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, "/root/reference")
import make_scene_warp_golden as _emu  # noqa: E402,F401  (the Warp stand-in + module stubs)
import warp as wp  # noqa: E402
from curobo._src.cost.wp_tool_pose import create_goalset_pose_distance_kernel_with_constants  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
unit = lambda q: q / np.linalg.norm(q, axis=-1, keepdims=True)  # noqa: E731
KEYS = ("distance", "position_distance", "rotation_distance", "position_gradient", "rotation_gradient")
bad = 0
kernels = {}
for case in range(n_cases):
    B, H, L, NG, G = int(rng.integers(1, 6)), int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 4))
    method = int(rng.integers(0, 3))
    goal_p = rng.uniform(-0.6, 0.6, (G, L, NG, 3)).astype(np.float32)
    goal_q = unit(rng.standard_normal((G, L, NG, 4))).astype(np.float32)
    idxs = rng.integers(0, G, size=B).astype(np.int32)
    cur_p = rng.uniform(-0.6, 0.6, (B, H, L, 3)).astype(np.float32)
    cur_q = unit(rng.standard_normal((B, H, L, 4))).astype(np.float32)
    for b in range(B):  # some poses near / at a goal member, on either hemisphere
        for h in range(H):
            u = rng.random()
            if u < 0.4:
                m = int(rng.integers(NG))
                eps = float(rng.choice([0.0, 1e-4, 0.02]))
                cur_p[b, h] = goal_p[idxs[b], :, m] + eps * rng.standard_normal((L, 3))
                dq = unit(np.concatenate([np.ones((L, 1)), eps * 2.5 * rng.standard_normal((L, 3))], -1))
                qg = goal_q[idxs[b], :, m].astype(np.float64)
                w1, v1, w2, v2 = qg[:, :1], qg[:, 1:], dq[:, :1], dq[:, 1:]
                comp = np.concatenate([w1 * w2 - (v1 * v2).sum(-1, keepdims=True), w1 * v2 + w2 * v1 + np.cross(v1, v2)], -1)
                cur_q[b, h] = (comp * (-1.0 if rng.random() < 0.5 else 1.0)).astype(np.float32)
    pw = rng.uniform(1.0, 50.0, size=2).astype(np.float32)
    axes = lambda: (rng.uniform(0.0, 2.0, (L, 6)) * (rng.random((L, 6)) > 0.2)).astype(np.float32)  # noqa: E731
    term_w, nonterm_w = axes(), axes() * float(rng.choice([0.0, 1.0]))
    term_tol = rng.choice([0.0, 0.001, 0.01], size=(L, 2)).astype(np.float32)
    nonterm_tol = rng.choice([0.0, 0.002, 0.05], size=(L, 2)).astype(np.float32)
    project = (rng.random((L, 1)) < 0.5).astype(np.uint8)
    n = B * H * L
    try:
        if (NG, method) not in kernels:
            kernels[(NG, method)] = create_goalset_pose_distance_kernel_with_constants(NG, method)
        o_dist, o_pd, o_rd = np.zeros(n * 2, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        o_pg, o_rg, o_idx = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.int32)
        wp.launch(kernel=kernels[(NG, method)], dim=n, inputs=[
            wp.array(cur_p.reshape(-1, 3), dtype=wp.vec3), wp.array(cur_q.reshape(-1, 4), dtype=wp.vec4),
            wp.array(goal_p.reshape(-1, 3), dtype=wp.vec3), wp.array(goal_q.reshape(-1, 4), dtype=wp.vec4),
            wp.array(idxs.reshape(-1), dtype=wp.int32), wp.array(pw), wp.array(term_w.reshape(-1)), wp.array(nonterm_w.reshape(-1)),
            wp.array(term_tol.reshape(-1)), wp.array(nonterm_tol.reshape(-1)), wp.array(project.reshape(-1), dtype=wp.uint8),
            wp.array(o_dist), wp.array(o_pd), wp.array(o_rd), wp.array(o_pg, dtype=wp.vec3), wp.array(o_rg, dtype=wp.vec4),
            wp.array(o_idx, dtype=wp.int32), B, H, L])
        want = dict(distance=o_dist.reshape(B, H, 2 * L), position_distance=o_pd.reshape(B, H, L), rotation_distance=o_rd.reshape(B, H, L),
                    position_gradient=o_pg.reshape(B, H, L, 3), rotation_gradient=o_rg.reshape(B, H, L, 4), goalset_idx=o_idx.reshape(B, H, L))
        r = oracle.tool_pose_distance(cur_p, cur_q, goal_p, goal_q, idxs, pw, term_w, nonterm_w, term_tol, nonterm_tol, project.reshape(-1),
                                      rotation_method=method)
        # the member picked: identical unless two members tie to rounding (then both costs agree anyway)
        pick_off = r["goalset_idx"] != want["goalset_idx"]
        for key in KEYS:
            got, w_ = r[key], want[key]
            np.testing.assert_allclose(got, w_, rtol=0, atol=1e-5 * max(1.0, float(np.abs(w_).max())), err_msg=f"{key} (method {method})")  # (unit gradients of poses 1e-4 from their goal carry the rounding of that difference)
        assert pick_off.mean() < 0.02, f"goal-set member picked differs on {int(pick_off.sum())} of {pick_off.size} poses"
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: B {B} H {H} L {L} NG {NG} G {G} method {method}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
