"""Tool-pose and c-space costs against the REFERENCE's own Warp kernels.

``tests/golden/tool_pose_warp_golden.npz`` / ``cspace_warp_golden.npz``: inputs and outputs of the reference's unmodified
``goalset_pose_distance`` kernel (rotation methods 0 / 1 / 2, goal sets, goal-frame projection, tolerances),
``forward_cspace_state_warp`` and ``forward_cspace_position_warp`` (+ ``warp_bound_util``), executed on the CPU through
the Warp stand-in of ``tests/golden/warp_emulator`` (generator: ``tests/golden/make_cost_warp_golden.py``).

CPU: the C oracle reproduces them (c-space: to the last bit but for pow(dt, 3); tool pose: ~2e-7 relative); GPU: the HIP
cost kernels through the C ABI at the path's 1e-5.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSE_KEYS = ("distance", "position_distance", "rotation_distance", "position_gradient", "rotation_gradient")
STATE_KEYS = ("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")
LIMITS = ("position", "velocity", "acceleration", "jerk", "effort")


def _close(got, want, tol, what):
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max())), err_msg=what)


# ---------------------------------------------------------------- tool pose
@pytest.mark.parametrize("method", [0, 1, 2])
def test_oracle_reproduces_the_reference_tool_pose_kernel(method, oracle):
    g = np.load(os.path.join(GOLD, "tool_pose_warp_golden.npz"))
    r = oracle.tool_pose_distance(g["current_position"], g["current_quat"], g["goal_position"], g["goal_quat"], g["idxs_goal"].reshape(-1),
                                  g["position_orientation_weight"], g["terminal_axes_weight"], g["non_terminal_axes_weight"],
                                  g["terminal_tolerance"], g["non_terminal_tolerance"], g["project_distance_to_goal"].reshape(-1),
                                  rotation_method=method)
    k = f"method{method}/"
    assert np.array_equal(r["goalset_idx"], g[k + "goalset_idx"])
    want = g[k + "distance"]
    assert 5 <= (want == 0).sum() < want.size // 2  # poses inside the tolerance are in the set, most are not
    assert np.array_equal(np.abs(r["distance"]) <= 1e-9, want == 0)  # (an exact match is 0 there and 1e-16 here: product order)
    for key in POSE_KEYS:
        _close(r[key], g[k + key], 1e-6, f"method {method} {key}")


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_hip_reproduces_the_reference_tool_pose_kernel(method, device):
    import torch

    from curobo_amd.backends import cost as Cs

    g = np.load(os.path.join(GOLD, "tool_pose_warp_golden.npz"))
    b, h, L, _ = g["current_position"].shape
    ng = g["goal_position"].shape[-2]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    out = dict(distance=torch.zeros(b, h, 2 * L, device=device), position_distance=torch.zeros(b, h, L, device=device),
               rotation_distance=torch.zeros(b, h, L, device=device), position_gradient=torch.zeros(b, h, L, 3, device=device),
               rotation_gradient=torch.zeros(b, h, L, 4, device=device),
               goalset_idx=torch.zeros(b, h, L, dtype=torch.int32, device=device))
    Cs.tool_pose_distance(out["distance"], out["position_distance"], out["rotation_distance"], out["position_gradient"],
                          out["rotation_gradient"], out["goalset_idx"], t(g["current_position"]), t(g["current_quat"]),
                          t(g["goal_position"]), t(g["goal_quat"]), t(g["idxs_goal"].reshape(-1)), t(g["position_orientation_weight"]),
                          t(g["terminal_axes_weight"]), t(g["non_terminal_axes_weight"]), t(g["terminal_tolerance"]),
                          t(g["non_terminal_tolerance"]), t(g["project_distance_to_goal"].reshape(-1)), b, h, L, ng, method)
    torch.cuda.synchronize()
    k = f"method{method}/"
    assert np.array_equal(out["goalset_idx"].cpu().numpy(), g[k + "goalset_idx"])
    for key in POSE_KEYS:
        _close(out[key].cpu().numpy(), g[k + key], 1e-5, f"method {method} {key}")


# ---------------------------------------------------------------- c-space
def _state_cases():
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    for name, prm in zip([str(x) for x in g["state_case_names"]], g["state_case_params"]):
        yield name, prm[0:5], prm[5:10], prm[10:15], float(prm[15]), float(prm[16]), bool(prm[17]), bool(prm[18])


def _position_cases():
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    for name, prm in zip([str(x) for x in g["position_case_names"]], g["position_case_params"]):
        yield name, prm[0:2], prm[2:4], float(prm[4]), prm[5:7], prm[7:9]


STATE_CASES, POSITION_CASES = list(_state_cases()), list(_position_cases())


@pytest.mark.parametrize("case", STATE_CASES, ids=[c[0] for c in STATE_CASES])
def test_oracle_reproduces_the_reference_cspace_state_kernel(case, oracle):
    name, w, act, reg, tw, ntf, rt, rtr = case
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    lim = {k: g["limit_" + k] for k in LIMITS}
    r = oracle.cspace_state_cost(g["pos"], g["vel"], g["acc"], g["jerk"], g["state_dt"], lim, w, act, reg, effort=g["effort"],
                                 target=g["target"], idxs_target=g["idxs_target"], target_weight=tw, non_terminal_factor=ntf,
                                 target_dof_weight=g["target_dof_weight"], retime_weights=rt, retime_regularization_weights=rtr)
    assert (g[name + "/cost"] != 0).sum() > 100
    for key in STATE_KEYS:
        _close(r[key], g[f"{name}/{key}"], 1e-6, f"{name} {key}")


@pytest.mark.parametrize("case", POSITION_CASES, ids=[c[0] for c in POSITION_CASES])
def test_oracle_reproduces_the_reference_cspace_position_kernel(case, oracle):
    name, w, act, tw, reg, dts = case
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    r = oracle.cspace_position_cost(g["position_pos"], g["limit_position"], w, act, effort=g["effort"], effort_b=g["limit_effort"],
                                    cspace_target=g["target"], cspace_target_idx=g["idxs_target"], cspace_target_weight=tw,
                                    cspace_target_dof_weight=g["target_dof_weight"], squared_l2_reg_weight=reg,
                                    current_position=g["position_current_position"], current_velocity=g["position_current_velocity"],
                                    idxs_current_state=g["position_idxs_current_state"], v_b=g["limit_velocity"], state_dt=dts)
    assert (g[name + "/cost"] != 0).sum() > 20
    for key in ("cost", "grad_position", "grad_effort"):
        _close(r[key], g[f"{name}/{key}"], 1e-6, f"{name} {key}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", STATE_CASES, ids=[c[0] for c in STATE_CASES])
def test_hip_reproduces_the_reference_cspace_state_kernel(case, device):
    import torch

    from curobo_amd.backends import cost as Cs

    name, w, act, reg, tw, ntf, rt, rtr = case
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    b, h, d = g["pos"].shape
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)  # noqa: E731
    outs = [torch.zeros(b, h, d, device=device) for _ in range(6)]
    Cs.cspace_state_cost(*outs, f(g["pos"]), f(g["vel"]), f(g["acc"]), f(g["jerk"]), f(g["effort"]), f(g["state_dt"]), f(g["target"]),
                         torch.as_tensor(g["idxs_target"].astype(np.int32), device=device), *[f(g["limit_" + k]) for k in LIMITS],
                         f(w), f(act), f(reg), f([tw]), f([ntf]), f(g["target_dof_weight"]), True, b, h, d, rt, rtr)
    torch.cuda.synchronize()
    for o, key in zip(outs, STATE_KEYS):
        _close(o.cpu().numpy(), g[f"{name}/{key}"], 1e-5, f"{name} {key}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", POSITION_CASES, ids=[c[0] for c in POSITION_CASES])
def test_hip_reproduces_the_reference_cspace_position_kernel(case, device):
    import torch

    from curobo_amd.backends import cost as Cs

    name, w, act, tw, reg, dts = case
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    b, h, d = g["position_pos"].shape
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)  # noqa: E731
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=device)  # noqa: E731
    oc, og, ot = (torch.zeros(b, h, d, device=device) for _ in range(3))
    Cs.cspace_position_cost(oc, og, ot, f(g["position_pos"]), f(g["effort"]), f(g["target"]), i32(g["idxs_target"]), f(g["limit_position"]),
                            f(g["limit_effort"]), f(w), f(act), f([tw]), f(g["target_dof_weight"]), f(reg), f(g["position_current_position"]),
                            f(g["position_current_velocity"]), i32(g["position_idxs_current_state"]), f(g["limit_velocity"]), f(dts),
                            True, b, h, d)
    torch.cuda.synchronize()
    for o, key in zip((oc, og, ot), ("cost", "grad_position", "grad_effort")):
        _close(o.cpu().numpy(), g[f"{name}/{key}"], 1e-5, f"{name} {key}")


def test_reference_l2_distance_kernel_golden_is_the_closed_form():
    """``forward_l2_warp`` (cost/wp_torch_cspace_dist.py) through the stand-in: w r_d (q - target)^2 with the terminal
    weights on the last point, entries of zero weight left untouched"""
    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    b, h, d = g["pos"].shape
    r = np.broadcast_to(g["l2_non_terminal_dof_weight"], (b, h, d)).copy()
    r[:, -1] = g["l2_terminal_dof_weight"]
    w = np.float32(g["l2_weight"]) * r
    err = g["pos"] - g["target"][g["idxs_target"]][:, None, :]
    assert np.array_equal(g["l2/cost"] == -7.0, w == 0) and np.array_equal(g["l2/grad_position"] == -7.0, w == 0)
    m = w != 0
    np.testing.assert_allclose(g["l2/cost"][m], (w * err * err)[m], rtol=1e-6)
    np.testing.assert_allclose(g["l2/grad_position"][m], (2 * w * err)[m], rtol=1e-6)


@pytest.mark.gpu
def test_hip_reproduces_the_reference_l2_distance_kernel(device):
    import torch

    from curobo_amd.backends import cost as Cs

    g = np.load(os.path.join(GOLD, "cspace_warp_golden.npz"))
    b, h, d = g["pos"].shape
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)  # noqa: E731
    oc, og = torch.full((b, h, d), -7.0, device=device), torch.full((b, h, d), -7.0, device=device)
    Cs.cspace_l2_distance(oc, og, f(g["pos"]), f(g["target"]), torch.as_tensor(g["idxs_target"].astype(np.int32), device=device),
                          f([g["l2_weight"]]), f(g["l2_terminal_dof_weight"]), f(g["l2_non_terminal_dof_weight"]), True, b, h, d)
    torch.cuda.synchronize()
    _close(oc.cpu().numpy(), g["l2/cost"], 1e-6, "l2 cost")
    _close(og.cpu().numpy(), g["l2/grad_position"], 1e-6, "l2 grad")


# ---------------------------------------------------------------- Levenberg-Marquardt step (Warp tile kernel)
LM_TAGS = ("ik13x7", "r20x6")


@pytest.mark.parametrize("tag", LM_TAGS)
def test_oracle_reproduces_the_reference_lm_tile_kernel(tag, oracle):
    """``LevenbergMarquardtStep.create_lm_warp_kernel`` run through the stand-in's tile API (generator
    ``tests/golden/make_lm_warp_golden.py``): the step, q_out = q_in + delta, and the predicted reduction
    0.5 * delta . (lambda delta - J^T e) -- same fp32 arithmetic in index order, so the oracle agrees to the last bit"""
    g = np.load(os.path.join(GOLD, "lm_warp_golden.npz"))
    q, pred = oracle.lm_step(g[tag + "/jacobian"], g[tag + "/jTerror"], g[tag + "/lambda"], g[tag + "/joint_position_in"])
    _close(q, g[tag + "/joint_position_out"], 1e-6, tag + " joint_position_out")
    _close(pred, g[tag + "/pred_reduction"], 1e-6, tag + " pred_reduction")
    assert np.all(g[tag + "/pred_reduction"] > 0)  # a descent step of the damped model


@pytest.mark.gpu
@pytest.mark.parametrize("tag", LM_TAGS)
def test_hip_reproduces_the_reference_lm_tile_kernel(tag, device):
    """the MFMA kernel sums J^T J in tile order, not index order: same tolerances as its test against the oracle"""
    import torch

    from curobo_amd.backends import linalg as La

    g = np.load(os.path.join(GOLD, "lm_warp_golden.npz"))
    J, jte, lam, q = (g[f"{tag}/{k}"] for k in ("jacobian", "jTerror", "lambda", "joint_position_in"))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    q_out, pred = torch.zeros(q.shape, device=device), torch.zeros(q.shape[0], device=device)
    La.levenberg_marquardt_step(q_out, pred, t(J), t(jte), t(lam), t(q))
    torch.cuda.synchronize()
    d_ref = g[tag + "/joint_position_out"] - q
    np.testing.assert_allclose(q_out.cpu().numpy() - q, d_ref, rtol=2e-3, atol=5e-4 * np.abs(d_ref).max())
    np.testing.assert_allclose(pred.cpu().numpy(), g[tag + "/pred_reduction"], rtol=2e-3, atol=1e-4 * np.abs(g[tag + "/pred_reduction"]).max())


@pytest.mark.skipif(not os.path.isdir("/root/reference/curobo/_src/cost"), reason="the reference's Warp sources are not on this machine")
@pytest.mark.parametrize("script,cases", [("sweep_reference_warp_tool_pose.py", "60"), ("sweep_reference_warp_scene.py", "40"), ("sweep_reference_warp_cspace.py", "40"), ("sweep_reference_warp_lm.py", "40")])
def test_randomised_sweep_against_the_reference_warp_kernels(script, cases):
    """the oracle against the reference's Warp kernels (through tests/golden/warp_emulator) on random inputs:
    tests/randomised/sweep_reference_warp_*.py at a small size"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", script), cases, "9"], capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]
