"""The ORACLE's c-space STATE and POSITION costs against the reference's own Warp kernels (wp_cspace_state.py,
wp_cspace_position.py, through tests/golden/warp_emulator) on RANDOM inputs: batch / horizon / dof, states inside / at /
beyond every limit, activation distances, weights with zeros, targets, per-dof weights, regularisation, retimed weights,
dt = 0.  CPU only, needs /root/reference.   python tests/randomised/sweep_reference_warp_cspace.py [cases] [seed]"""
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
import make_scene_warp_golden as _emu  # noqa: E402,F401
import warp as wp  # noqa: E402
from curobo._src.cost.wp_cspace_position import forward_cspace_position_warp  # noqa: E402
from curobo._src.cost.wp_cspace_state import forward_cspace_state_warp  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LIMITS = ("position", "velocity", "acceleration", "jerk", "effort")
STATE_KEYS = ("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")
f = lambda a: wp.array(np.ascontiguousarray(a, np.float32).reshape(-1))  # noqa: E731
bad = 0


def close(got, want, what):
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * max(1.0, float(np.abs(want).max())), err_msg=what)


for case in range(n_cases):
    B, H, D = int(rng.integers(1, 6)), int(rng.integers(1, 8)), int(rng.integers(1, 10))
    n = B * H * D
    lim = {k: np.stack([-(a + b * rng.random(D)), a + b * rng.random(D)]).astype(np.float32)
           for k, (a, b) in dict(position=(1.5, 1.0), velocity=(1.0, 1.5), acceleration=(5.0, 5.0), jerk=(50.0, 50.0), effort=(20.0, 60.0)).items()}
    around = lambda key: (lim[key][1] * rng.uniform(-1.1, 1.1, (B, H, D))).astype(np.float32)  # noqa: E731
    pos, vel, acc, jerk, eff = (around(k) for k in LIMITS)
    if rng.random() < 0.5:
        pos[0, 0] = lim["position"][1]  # exactly on a limit
    if rng.random() < 0.5:
        vel[rng.random((B, H, D)) < 0.3] = 0.0
    state_dt = (0.02 + 0.2 * rng.random(B)).astype(np.float32)
    nt = int(rng.integers(1, 4))
    target = rng.uniform(-1, 1, (nt, D)).astype(np.float32)
    idxs_target = rng.integers(0, nt, size=B).astype(np.int32)
    dof_w = (rng.uniform(0, 2, D) * (rng.random(D) > 0.2)).astype(np.float32)
    w = (rng.choice([0.0, 1.0, 50.0, 5000.0], size=5)).astype(np.float32)
    act = rng.choice([0.0, 0.02, 0.1], size=5).astype(np.float32)
    reg = (rng.choice([0.0, 0.001, 0.3], size=5)).astype(np.float32)
    tw, ntf = float(rng.choice([0.0, 2.0, 7.5])), float(rng.choice([0.0, 0.2, 1.0]))
    rt, rtr = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    try:
        o = [np.zeros(n, np.float32) for _ in range(6)]
        wp.launch(kernel=forward_cspace_state_warp, dim=n, inputs=[
            f(pos), f(vel), f(acc), f(jerk), f(eff), f(state_dt), f(target), wp.array(idxs_target, dtype=wp.int32),
            f(lim["position"]), f(lim["velocity"]), f(lim["acceleration"]), f(lim["jerk"]), f(lim["effort"]), f(w), f(act), f(reg),
            f([tw]), f([ntf]), f(dof_w), *[wp.array(x) for x in o], wp.uint8(1), B, H, D, rt, rtr])
        r = oracle.cspace_state_cost(pos, vel, acc, jerk, state_dt, lim, w, act, reg, effort=eff, target=target, idxs_target=idxs_target,
                                     target_weight=tw, non_terminal_factor=ntf, target_dof_weight=dof_w, retime_weights=rt,
                                     retime_regularization_weights=rtr)
        for key, arr in zip(STATE_KEYS, o):
            close(r[key], arr.reshape(B, H, D), f"STATE {key}")
        # ---- POSITION kernel
        nc = int(rng.integers(1, 3))
        cur_p, cur_v = rng.uniform(-1, 1, (nc, D)).astype(np.float32), rng.uniform(-1, 1, (nc, D)).astype(np.float32)
        idxs_cur = rng.integers(0, nc, size=B).astype(np.int32)
        pos2 = (cur_p[idxs_cur][:, None, :] + 0.4 * rng.standard_normal((B, H, D))).astype(np.float32)
        pos2[:, ::2] = pos[:, ::2]
        w2, act2 = rng.choice([0.0, 3.0, 400.0], size=2).astype(np.float32), rng.choice([0.0, 0.05, 0.1], size=2).astype(np.float32)
        reg2 = rng.choice([0.0, 0.004, 0.2], size=2).astype(np.float32)
        dts = rng.choice([0.0, 0.05, 0.11], size=2).astype(np.float32)
        tw2 = float(rng.choice([0.0, 1.5]))
        o = [np.zeros(n, np.float32) for _ in range(3)]
        wp.launch(kernel=forward_cspace_position_warp, dim=n, inputs=[
            f(pos2), f(eff), f(target), wp.array(idxs_target, dtype=wp.int32), f(lim["position"]), f(lim["effort"]), f(w2), f(act2), f([tw2]),
            f(dof_w), f(reg2), f(cur_p), f(cur_v), wp.array(idxs_cur, dtype=wp.int32), f(lim["velocity"]), f(dts),
            *[wp.array(x) for x in o], wp.uint8(1), B, H, D])
        r = oracle.cspace_position_cost(pos2, lim["position"], w2, act2, effort=eff, effort_b=lim["effort"], cspace_target=target,
                                        cspace_target_idx=idxs_target, cspace_target_weight=tw2, cspace_target_dof_weight=dof_w,
                                        squared_l2_reg_weight=reg2, current_position=cur_p, current_velocity=cur_v, idxs_current_state=idxs_cur,
                                        v_b=lim["velocity"], state_dt=dts)
        for key, arr in zip(("cost", "grad_position", "grad_effort"), o):
            close(r[key], arr.reshape(B, H, D), f"POSITION {key}")
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: B {B} H {H} D {D} w {w.tolist()} act {act.tolist()} retime {rt} {rtr}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
