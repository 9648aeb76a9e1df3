"""Golden vectors for the tool-pose (goal-set) cost, produced by the REFERENCE's own Warp kernel executed on the CPU
through the Warp stand-in of tests/golden/warp_emulator (see its docstring and make_scene_warp_golden.py):

    PYTHONPATH=/root/reference python tests/golden/make_cost_warp_golden.py

    curobo/_src/cost/wp_tool_pose.py   create_goalset_pose_distance_kernel_with_constants(num_goalset, rotation_method)
                                       for rotation_method 0 (axis-angle), 1 (Lie group), 2 (Lie group, advanced), launched
                                       as ToolPoseDistance.forward does (:806-836): dim = batch * horizon * links

Cases: goal sets of three poses, two links (one measured in the world frame, one projected into the goal frame),
terminal / non-terminal weights and tolerances, goals shared through idxs_goal, current orientations on both
quaternion hemispheres, poses already inside the convergence tolerance (zero cost and gradient), an exact match.
Output: tests/golden/tool_pose_warp_golden.npz.

    curobo/_src/cost/wp_cspace_state.py      forward_cspace_state_warp (+ warp_bound_util.py), launched as
                                             StateCSpaceFunction.forward does: dim = batch * horizon * dof
    curobo/_src/cost/wp_cspace_position.py   forward_cspace_position_warp, as PositionCSpaceFunction.forward

Cases: states inside / at / beyond every limit (position, velocity, acceleration, jerk, effort), activation distances
(fractions of the range),
joint-position targets with per-dof weights and the non-terminal factor, squared-L2 and energy regularisation, weights
retimed by the per-trajectory dt; for the position kernel: velocity-clamped bounds around a current state, implied
velocity / acceleration regularisation, and dt = 0 (both off).  Output: tests/golden/cspace_warp_golden.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_scene_warp_golden as _emu  # noqa: E402,F401  (puts the emulator + module stubs in place)

import warp as wp  # noqa: E402

from curobo._src.cost.wp_cspace_position import forward_cspace_position_warp  # noqa: E402
from curobo._src.cost.wp_cspace_state import forward_cspace_state_warp  # noqa: E402
from curobo._src.cost.wp_tool_pose import create_goalset_pose_distance_kernel_with_constants  # noqa: E402
from curobo._src.cost.wp_torch_cspace_dist import forward_l2_warp  # noqa: E402


def unit(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def main():
    rng = np.random.default_rng(77)
    B, H, L, NG, G = 6, 4, 2, 3, 3
    goal_p = rng.uniform(-0.6, 0.6, (G, L, NG, 3)).astype(np.float32)
    goal_q = unit(rng.standard_normal((G, L, NG, 4))).astype(np.float32)  # w, x, y, z
    idxs_goal = np.array([0, 1, 2, 1, 0, 2], np.int32).reshape(B, 1)
    cur_p = rng.uniform(-0.6, 0.6, (B, H, L, 3)).astype(np.float32)
    cur_q = unit(rng.standard_normal((B, H, L, 4))).astype(np.float32)
    # near the goals (small errors, both hemispheres), inside the tolerance, and an exact match
    for b in range(B):
        g = idxs_goal[b, 0]
        cur_p[b, 1] = goal_p[g, :, 1] + 0.02 * rng.standard_normal((L, 3))
        dq = unit(np.concatenate([np.ones((L, 1)), 0.05 * rng.standard_normal((L, 3))], -1))
        qg = goal_q[g, :, 1].astype(np.float64)
        w1, v1, w2, v2 = qg[:, :1], qg[:, 1:], dq[:, :1], dq[:, 1:]
        comp = np.concatenate([w1 * w2 - (v1 * v2).sum(-1, keepdims=True), w1 * v2 + w2 * v1 + np.cross(v1, v2)], -1)
        cur_q[b, 1] = (comp * (-1.0 if b % 2 else 1.0)).astype(np.float32)
    cur_p[2, 2] = goal_p[idxs_goal[2, 0], :, 0] + 1e-4
    cur_q[2, 2] = goal_q[idxs_goal[2, 0], :, 0]
    cur_p[3, 3] = goal_p[idxs_goal[3, 0], :, 2]
    cur_q[3, 3] = goal_q[idxs_goal[3, 0], :, 2]
    pw = np.array([35.0, 12.0], np.float32)
    term_w = np.array([[1.0, 1.0, 1.0, 1.0, 1.0, 1.0], [1.0, 0.5, 2.0, 0.3, 1.0, 0.0]], np.float32)
    nonterm_w = np.array([[0.2, 0.2, 0.2, 0.1, 0.1, 0.1], [0.0, 1.0, 1.0, 1.0, 0.0, 1.0]], np.float32)
    term_tol = np.array([[0.001, 0.01], [0.0, 0.0]], np.float32)
    nonterm_tol = np.array([[0.01, 0.05], [0.002, 0.02]], np.float32)
    project = np.array([[0], [1]], np.uint8)
    out = dict(current_position=cur_p, current_quat=cur_q, goal_position=goal_p, goal_quat=goal_q, idxs_goal=idxs_goal,
               position_orientation_weight=pw, terminal_axes_weight=term_w, non_terminal_axes_weight=nonterm_w,
               terminal_tolerance=term_tol, non_terminal_tolerance=nonterm_tol, project_distance_to_goal=project)
    n = B * H * L
    for method in (0, 1, 2):
        kern = create_goalset_pose_distance_kernel_with_constants(NG, method)
        o_dist, o_pd, o_rd = np.zeros(n * 2, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        o_pg, o_rg, o_idx = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.int32)
        wp.launch(kernel=kern, dim=n, inputs=[
            wp.array(cur_p.reshape(-1, 3), dtype=wp.vec3), wp.array(cur_q.reshape(-1, 4), dtype=wp.vec4),
            wp.array(goal_p.reshape(-1, 3), dtype=wp.vec3), wp.array(goal_q.reshape(-1, 4), dtype=wp.vec4),
            wp.array(idxs_goal.reshape(-1), dtype=wp.int32), wp.array(pw), wp.array(term_w.reshape(-1)),
            wp.array(nonterm_w.reshape(-1)), wp.array(term_tol.reshape(-1)), wp.array(nonterm_tol.reshape(-1)),
            wp.array(project.reshape(-1), dtype=wp.uint8), wp.array(o_dist), wp.array(o_pd), wp.array(o_rd),
            wp.array(o_pg, dtype=wp.vec3), wp.array(o_rg, dtype=wp.vec4), wp.array(o_idx, dtype=wp.int32), B, H, L])
        k = f"method{method}"
        out[f"{k}/distance"] = o_dist.reshape(B, H, 2 * L)
        out[f"{k}/position_distance"], out[f"{k}/rotation_distance"] = o_pd.reshape(B, H, L), o_rd.reshape(B, H, L)
        out[f"{k}/position_gradient"], out[f"{k}/rotation_gradient"] = o_pg.reshape(B, H, L, 3), o_rg.reshape(B, H, L, 4)
        out[f"{k}/goalset_idx"] = o_idx.reshape(B, H, L)
        print(k, "zero costs", int((out[f"{k}/distance"] == 0).sum()), "of", 2 * n, "max", float(o_dist.max()),
              "goal picks", np.bincount(o_idx, minlength=NG).tolist())
    path = os.path.join(HERE, "tool_pose_warp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


def cspace():
    rng = np.random.default_rng(78)
    B, H, D = 5, 6, 7
    n = B * H * D
    lim = {k: np.stack([-(a + b * rng.random(D)), a + b * rng.random(D)]).astype(np.float32)
           for k, (a, b) in dict(position=(1.5, 1.0), velocity=(1.0, 1.5), acceleration=(5.0, 5.0), jerk=(50.0, 50.0), effort=(20.0, 60.0)).items()}

    def around(key, scale):  # values that straddle the limits of `key`
        hi = lim[key][1]
        return (scale * hi * rng.uniform(-1.08, 1.08, (B, H, D))).astype(np.float32)

    pos, vel, acc, jerk, eff = around("position", 1.0), around("velocity", 1.0), around("acceleration", 1.0), around("jerk", 1.0), around("effort", 1.0)
    pos[0, 0] = lim["position"][1]            # exactly on the upper limit
    vel[1, 2] = 0.0
    state_dt = (0.02 + 0.2 * rng.random(B)).astype(np.float32)
    target = rng.uniform(-1, 1, (3, D)).astype(np.float32)
    idxs_target = np.array([0, 2, 1, 1, 0], np.int32)
    dof_w = np.array([1.0, 0.5, 0.0, 2.0, 1.0, 1.0, 0.25], np.float32)
    out = dict(pos=pos, vel=vel, acc=acc, jerk=jerk, effort=eff, state_dt=state_dt, target=target, idxs_target=idxs_target,
               target_dof_weight=dof_w, **{f"limit_{k}": v for k, v in lim.items()})
    state_cases = [  # name, weight[5], activation[5], sql2[5], target_w, non_terminal_factor, retime, retime_reg
        ("state_plain", [5000.0, 500.0, 50.0, 5.0, 10.0], [0.02, 0.05, 0.1, 0.1, 0.05], [0.0, 0.0, 0.0, 0.0, 0.0], 0.0, 1.0, False, False),
        ("state_target_reg", [100.0, 10.0, 1.0, 0.1, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0], [0.3, 0.02, 0.001, 0.004, 0.0], 7.5, 0.2, False, False),
        ("state_retimed_energy", [300.0, 30.0, 3.0, 0.3, 1.0], [0.01, 0.02, 0.05, 0.05, 0.02], [0.1, 0.01, 0.0005, 0.002, 0.05], 2.0, 1.0, True, True),
    ]
    meta = []
    for name, w, act, reg, tw, ntf, rt, rtr in state_cases:
        o = [np.zeros(n, np.float32) for _ in range(6)]
        f = lambda a: wp.array(np.ascontiguousarray(a, np.float32).reshape(-1))  # noqa: E731
        wp.launch(kernel=forward_cspace_state_warp, dim=n, inputs=[
            f(pos), f(vel), f(acc), f(jerk), f(eff), f(state_dt), f(target), wp.array(idxs_target, dtype=wp.int32),
            f(lim["position"]), f(lim["velocity"]), f(lim["acceleration"]), f(lim["jerk"]), f(lim["effort"]), f(w), f(act), f(reg),
            f([tw]), f([ntf]), f(dof_w), *[wp.array(x) for x in o], wp.uint8(1), B, H, D, rt, rtr])
        for key, arr in zip(("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort"), o):
            out[f"{name}/{key}"] = arr.reshape(B, H, D)
        meta.append((name, *w, *act, *reg, tw, ntf, float(rt), float(rtr)))
        print(name, "violations", int((o[0] > 0).sum()), "of", n, "max cost", float(o[0].max()))
    out["state_case_names"] = np.array([m[0] for m in meta])
    out["state_case_params"] = np.array([m[1:] for m in meta], np.float64)  # weight[5], activation[5], sql2[5], target_w, factor, retime, retime_reg
    # ---- position kernel
    cur_p = rng.uniform(-1, 1, (2, D)).astype(np.float32)
    cur_v = rng.uniform(-1, 1, (2, D)).astype(np.float32)
    idxs_cur = np.array([0, 1, 1, 0, 1], np.int32)
    pos2 = (cur_p[idxs_cur][:, None, :] + 0.4 * rng.standard_normal((B, H, D))).astype(np.float32)
    pos2[:, ::2] = pos[:, ::2]  # every other point straddles the joint limits themselves
    out.update(position_pos=pos2, position_current_position=cur_p, position_current_velocity=cur_v, position_idxs_current_state=idxs_cur)
    pos_cases = [  # name, weight[2], activation[2], target_w, sql2[2], state_dt[2]
        ("position_clamped_reg", [400.0, 3.0], [0.05, 0.1], 1.5, [0.2, 0.004], [0.05, 0.11]),
        ("position_plain", [400.0, 0.0], [0.1, 0.0], 0.0, [0.0, 0.0], [0.0, 0.0]),
    ]
    meta = []
    for name, w, act, tw, reg, dts in pos_cases:
        o = [np.zeros(n, np.float32) for _ in range(3)]
        f = lambda a: wp.array(np.ascontiguousarray(a, np.float32).reshape(-1))  # noqa: E731
        wp.launch(kernel=forward_cspace_position_warp, dim=n, inputs=[
            f(pos2), f(eff), f(target), wp.array(idxs_target, dtype=wp.int32), f(lim["position"]), f(lim["effort"]), f(w), f(act), f([tw]),
            f(dof_w), f(reg), f(cur_p), f(cur_v), wp.array(idxs_cur, dtype=wp.int32), f(lim["velocity"]), f(dts),
            *[wp.array(x) for x in o], wp.uint8(1), B, H, D])
        for key, arr in zip(("cost", "grad_position", "grad_effort"), o):
            out[f"{name}/{key}"] = arr.reshape(B, H, D)
        meta.append((name, *w, *act, tw, *reg, *dts))
        print(name, "nonzero", int((o[0] > 0).sum()), "of", n, "max cost", float(o[0].max()))
    out["position_case_names"] = np.array([m[0] for m in meta])
    out["position_case_params"] = np.array([m[1:] for m in meta], np.float64)  # weight[2], activation[2], target_w, sql2[2], state_dt[2]
    # ---- joint-space L2 distance to a target (cost/wp_torch_cspace_dist.py: forward_l2_warp, as L2DistFunction.forward)
    term_w = np.array([1.0, 0.5, 2.0, 0.0, 1.0, 1.0, 0.25], np.float32)
    nonterm_w = np.array([0.1, 0.0, 0.2, 0.3, 0.0, 0.1, 0.05], np.float32)
    o = [np.full(n, -7.0, np.float32) for _ in range(2)]  # entries of zero weight are not written
    f = lambda a: wp.array(np.ascontiguousarray(a, np.float32).reshape(-1))  # noqa: E731
    wp.launch(kernel=forward_l2_warp, dim=n, inputs=[f(pos), f(target), wp.array(idxs_target, dtype=wp.int32), f([3.5]), f(term_w), f(nonterm_w),
                                                      wp.array(o[0]), wp.array(o[1]), wp.uint8(1), B, H, D])
    out.update(l2_terminal_dof_weight=term_w, l2_non_terminal_dof_weight=nonterm_w, l2_weight=np.float32(3.5),
               **{"l2/cost": o[0].reshape(B, H, D), "l2/grad_position": o[1].reshape(B, H, D)})
    print("l2 untouched entries", int((o[0] == -7.0).sum()), "of", n)
    path = os.path.join(HERE, "cspace_warp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
    cspace()
