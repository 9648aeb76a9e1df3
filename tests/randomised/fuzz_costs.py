"""The HIP cost kernels (tool pose with goal sets, c-space STATE, c-space POSITION) against the oracle on random inputs -- the
same generators as the CPU sweeps of the oracle against the reference's Warp kernels (sweep_reference_warp_*.py).
    python tests/randomised/fuzz_costs.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from curobo_amd.backends import cost as Cs  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
unit = lambda q: q / np.linalg.norm(q, axis=-1, keepdims=True)  # noqa: E731
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)  # noqa: E731
i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev)  # noqa: E731
LIMITS = ("position", "velocity", "acceleration", "jerk", "effort")
STATE_KEYS = ("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")
POSE_KEYS = ("distance", "position_distance", "rotation_distance", "position_gradient", "rotation_gradient")
bad = 0


def close(got, want, what, tol=1e-5):
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max())), err_msg=what)


for case in range(n_cases):
    # ---------------- tool pose
    B, H, L, NG, G = int(rng.integers(1, 40)), int(rng.integers(1, 34)), int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 4))
    method = int(rng.integers(0, 3))
    goal_p = rng.uniform(-0.6, 0.6, (G, L, NG, 3)).astype(np.float32)
    goal_q = unit(rng.standard_normal((G, L, NG, 4))).astype(np.float32)
    idxs = rng.integers(0, G, size=B).astype(np.int32)
    cur_p = rng.uniform(-0.6, 0.6, (B, H, L, 3)).astype(np.float32)
    cur_q = unit(rng.standard_normal((B, H, L, 4))).astype(np.float32)
    near = rng.random((B, H)) < 0.3
    for b, h in zip(*np.nonzero(near)):
        m = int(rng.integers(NG))
        eps = float(rng.choice([0.0, 1e-3, 0.02]))
        cur_p[b, h] = goal_p[idxs[b], :, m] + eps * rng.standard_normal((L, 3))
        cur_q[b, h] = unit(goal_q[idxs[b], :, m] * (-1.0 if rng.random() < 0.5 else 1.0) + eps * rng.standard_normal((L, 4))).astype(np.float32)
    pw = rng.uniform(1.0, 50.0, size=2).astype(np.float32)
    axes = lambda: (rng.uniform(0.0, 2.0, (L, 6)) * (rng.random((L, 6)) > 0.2)).astype(np.float32)  # noqa: E731
    term_w, nonterm_w = axes(), axes() * float(rng.choice([0.0, 1.0]))
    term_tol = rng.choice([0.0, 0.001, 0.01], size=(L, 2)).astype(np.float32)
    nonterm_tol = rng.choice([0.0, 0.002, 0.05], size=(L, 2)).astype(np.float32)
    project = (rng.random(L) < 0.5).astype(np.uint8)
    try:
        r = oracle.tool_pose_distance(cur_p, cur_q, goal_p, goal_q, idxs, pw, term_w, nonterm_w, term_tol, nonterm_tol, project, rotation_method=method)
        out = dict(distance=torch.zeros(B, H, 2 * L, device=dev), position_distance=torch.zeros(B, H, L, device=dev),
                   rotation_distance=torch.zeros(B, H, L, device=dev), position_gradient=torch.zeros(B, H, L, 3, device=dev),
                   rotation_gradient=torch.zeros(B, H, L, 4, device=dev), goalset_idx=torch.zeros(B, H, L, dtype=torch.int32, device=dev))
        Cs.tool_pose_distance(out["distance"], out["position_distance"], out["rotation_distance"], out["position_gradient"], out["rotation_gradient"],
                              out["goalset_idx"], t(cur_p), t(cur_q), t(goal_p), t(goal_q), t(idxs), t(pw), t(term_w), t(nonterm_w), t(term_tol),
                              t(nonterm_tol), t(project), B, H, L, NG, method)
        torch.cuda.synchronize()
        pick_off = out["goalset_idx"].cpu().numpy() != r["goalset_idx"]
        assert pick_off.mean() < 0.02, f"goal-set member differs on {int(pick_off.sum())} of {pick_off.size}"
        for key in POSE_KEYS:
            got, want = out[key].cpu().numpy(), r[key]
            ok = ~np.broadcast_to(pick_off.reshape(pick_off.shape + (1,) * (want.ndim - 3)), want.shape) if key != "distance" else np.ones(want.shape, bool)
            close(got[ok], want[ok], f"tool pose {key} (method {method})", 2e-5)
    except AssertionError as e:
        bad += 1
        print(f"FAILED tool pose case {case}: B {B} H {H} L {L} NG {NG} method {method}: {str(e)[:300]}".replace("\n", " | "))
    # ---------------- c-space STATE / POSITION
    B, H, D = int(rng.integers(1, 40)), int(rng.integers(1, 34)), int(rng.integers(1, 50))
    lim = {k: np.stack([-(a + b * rng.random(D)), a + b * rng.random(D)]).astype(np.float32)
           for k, (a, b) in dict(position=(1.5, 1.0), velocity=(1.0, 1.5), acceleration=(5.0, 5.0), jerk=(50.0, 50.0), effort=(20.0, 60.0)).items()}
    around = lambda key: (lim[key][1] * rng.uniform(-1.1, 1.1, (B, H, D))).astype(np.float32)  # noqa: E731
    pos, vel, acc, jerk, eff = (around(k) for k in LIMITS)
    state_dt = (0.02 + 0.2 * rng.random(B)).astype(np.float32)
    nt = int(rng.integers(1, 4))
    target, idxs_target = rng.uniform(-1, 1, (nt, D)).astype(np.float32), rng.integers(0, nt, size=B).astype(np.int32)
    dof_w = (rng.uniform(0, 2, D) * (rng.random(D) > 0.2)).astype(np.float32)
    w, act = rng.choice([0.0, 1.0, 50.0, 5000.0], size=5).astype(np.float32), rng.choice([0.0, 0.02, 0.1], size=5).astype(np.float32)
    reg = rng.choice([0.0, 0.001, 0.3], size=5).astype(np.float32)
    tw, ntf = float(rng.choice([0.0, 2.0, 7.5])), float(rng.choice([0.0, 0.2, 1.0]))
    rt, rtr = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    try:
        r = oracle.cspace_state_cost(pos, vel, acc, jerk, state_dt, lim, w, act, reg, effort=eff, target=target, idxs_target=idxs_target,
                                     target_weight=tw, non_terminal_factor=ntf, target_dof_weight=dof_w, retime_weights=rt, retime_regularization_weights=rtr)
        outs = [torch.zeros(B, H, D, device=dev) for _ in range(6)]
        Cs.cspace_state_cost(*outs, f(pos), f(vel), f(acc), f(jerk), f(eff), f(state_dt), f(target), i32(idxs_target), *[f(lim[k]) for k in LIMITS],
                             f(w), f(act), f(reg), f([tw]), f([ntf]), f(dof_w), True, B, H, D, rt, rtr)
        torch.cuda.synchronize()
        for o, key in zip(outs, STATE_KEYS):
            close(o.cpu().numpy(), r[key], f"STATE {key}")
        nc = int(rng.integers(1, 3))
        cur_p2, cur_v2 = rng.uniform(-1, 1, (nc, D)).astype(np.float32), rng.uniform(-1, 1, (nc, D)).astype(np.float32)
        idxs_cur = rng.integers(0, nc, size=B).astype(np.int32)
        pos2 = (cur_p2[idxs_cur][:, None, :] + 0.4 * rng.standard_normal((B, H, D))).astype(np.float32)
        w2, act2 = rng.choice([0.0, 3.0, 400.0], size=2).astype(np.float32), rng.choice([0.0, 0.05, 0.1], size=2).astype(np.float32)
        reg2, dts, tw2 = rng.choice([0.0, 0.004, 0.2], size=2).astype(np.float32), rng.choice([0.0, 0.05, 0.11], size=2).astype(np.float32), float(rng.choice([0.0, 1.5]))
        r = oracle.cspace_position_cost(pos2, lim["position"], w2, act2, effort=eff, effort_b=lim["effort"], cspace_target=target, cspace_target_idx=idxs_target,
                                        cspace_target_weight=tw2, cspace_target_dof_weight=dof_w, squared_l2_reg_weight=reg2, current_position=cur_p2,
                                        current_velocity=cur_v2, idxs_current_state=idxs_cur, v_b=lim["velocity"], state_dt=dts)
        oc, og, ot = (torch.zeros(B, H, D, device=dev) for _ in range(3))
        Cs.cspace_position_cost(oc, og, ot, f(pos2), f(eff), f(target), i32(idxs_target), f(lim["position"]), f(lim["effort"]), f(w2), f(act2), f([tw2]),
                                f(dof_w), f(reg2), f(cur_p2), f(cur_v2), i32(idxs_cur), f(lim["velocity"]), f(dts), True, B, H, D)
        torch.cuda.synchronize()
        for o, key in zip((oc, og, ot), ("cost", "grad_position", "grad_effort")):
            close(o.cpu().numpy(), r[key], f"POSITION {key}")
    except AssertionError as e:
        bad += 1
        print(f"FAILED c-space case {case}: B {B} H {H} D {D}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
