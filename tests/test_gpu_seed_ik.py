"""Seed-IK (Levenberg-Marquardt) solver on the HIP kernels against the CPU restatement of the
reference's iteration (oracle/seed_ik_ref.py): state-update kernel, one evaluation, the first
iterations, and the solve statistics."""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def _problem(oracle, model, P, S, seed=0):
    md = model.as_dict()
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    qg = (lo + (hi - lo) * rng.random((P, lo.shape[0]))).astype(np.float32)
    fk = oracle.kinematics_forward(qg, md, compute_spheres=False)
    T = md["tool_frame_map"].shape[0]
    seeds = (lo + (hi - lo) * rng.random((P * S, lo.shape[0]))).astype(np.float32)
    return md, fk["link_pos"].reshape(P, T, 1, 3), fk["link_quat"].reshape(P, T, 1, 4), seeds, np.repeat(np.arange(P, dtype=np.int32), S)


def _solver(device, P, S, **kw):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.solver.seed_ik import SeedIKSolver, SeedIKSolverCfg

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    return model, SeedIKSolver(kin, P, SeedIKSolverCfg(num_seeds=S, **kw))


def test_update_state_kernel_matches_reference_logic(device):
    from curobo_amd.backends import linalg
    from oracle import seed_ik_ref as R

    rng = np.random.default_rng(3)
    n, D, T = 203, 7, 2
    Rr = 6 * T + D
    cfg = R.SeedIKRefCfg()
    lo, hi = -np.ones(D, np.float32), np.ones(D, np.float32)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    cur = {"joint_position": 0.7 * f(n, D), "jacobian": f(n, Rr, D), "jTerror": f(n, D), "error_norm": np.abs(f(n)) + 0.5,
           "position_errors": np.abs(f(n)) * 1e-5, "orientation_errors": np.abs(f(n)) * 1e-5,
           "lambda_damping": np.abs(f(n)) + 0.01}
    cq = 0.8 * f(n, D)
    pose_jac, pose_jte = f(n, 6 * T, D), f(n, D)
    pose_cost, pd, rd = np.abs(f(n, T, 2)) * 0.5, np.abs(f(n, T)) * 1e-5, np.abs(f(n, T)) * 1e-5
    pred = f(n) * 0.5
    pred[:5] = 0.0
    # candidate in the reference's form
    uv, lv = np.maximum(cq - hi, 0), np.maximum(lo - cq, 0)
    jl = cfg.joint_limit_weight * (lv + uv)
    diag = cfg.joint_limit_weight * (np.where(lv > 0, -1.0, 0.0) + np.where(uv > 0, 1.0, 0.0)).astype(np.float32)
    J = np.zeros((n, Rr, D), np.float32)
    J[:, :6 * T] = pose_jac
    J[:, 6 * T + np.arange(D), np.arange(D)] = diag
    cand = {"joint_position": cq, "jacobian": J, "jTerror": pose_jte + diag * jl,
            "error_norm": pose_cost.reshape(n, -1).sum(-1) + jl.sum(-1), "position_errors": pd.max(-1), "orientation_errors": rd.max(-1)}
    ref = R.update_state(cur, cand, pred, lo, hi, cfg)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dt)  # noqa: E731
    st = {k: t(v) for k, v in cur.items()}
    succ, imp = torch.zeros(n, dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.uint8, device=device)
    linalg.seed_ik_update_state(
        st["joint_position"], st["jacobian"], st["jTerror"], st["error_norm"], st["position_errors"], st["orientation_errors"],
        st["lambda_damping"], succ, imp, t(cq), t(pose_jac), t(pose_jte), t(pose_cost), t(pd), t(rd), t(pred), t(lo), t(hi),
        None, None, None, cfg.joint_limit_weight, cfg.rho_min, cfg.lambda_factor, cfg.lambda_min, cfg.lambda_max,
        cfg.convergence_position_tolerance, cfg.convergence_orientation_tolerance, cfg.convergence_joint_limit_weight, False)
    torch.cuda.synchronize()
    # decisions: exact, except where the trust ratio sits within rounding of the threshold
    rho = (cur["error_norm"] - cand["error_norm"]) / (pred + np.float32(1e-8))
    clear = np.abs(rho - cfg.rho_min) > 1e-4 * (1 + np.abs(rho))
    assert np.array_equal(imp.cpu().numpy().astype(bool)[clear], ref["improvement"][clear])
    m = clear
    assert np.array_equal(succ.cpu().numpy().astype(bool)[m], ref["success"][m])
    for k in ("joint_position", "jacobian", "jTerror", "position_errors", "orientation_errors", "lambda_damping"):
        np.testing.assert_allclose(st[k].cpu().numpy()[m], ref[k][m], rtol=1e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(st["error_norm"].cpu().numpy(), ref["error_norm"], rtol=1e-5, atol=1e-6)
    assert ref["improvement"].any() and (~ref["improvement"]).any() and ref["success"].any()


def test_evaluation_and_first_iterations_match_oracle(oracle, device):
    from oracle import seed_ik_ref as R

    P, S = 12, 6
    model, solver = _solver(device, P, S, use_cuda_graph=False, max_iterations=4, inner_iterations=1, batch_success_threshold=2.0)
    md, gp, gq, seeds, idx = _problem(oracle, model, P, S)
    cfg = R.SeedIKRefCfg(max_iterations=2)
    ref0 = R.evaluate(oracle, md, cfg, seeds, gp, gq, idx)
    solver.goal_position.copy_(torch.as_tensor(gp, device=device))
    solver.goal_quat.copy_(torch.as_tensor(gq, device=device))
    solver.lambda_damping.fill_(cfg.lambda_initial)
    solver._evaluate_candidate(torch.as_tensor(seeds, device=device), initial=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(solver.jacobian.cpu().numpy(), ref0["jacobian"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(solver.jTerror.cpu().numpy(), ref0["jTerror"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(solver.error_norm.cpu().numpy(), ref0["error_norm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(solver.position_error.cpu().numpy(), ref0["position_errors"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(solver.orientation_error.cpu().numpy(), ref0["orientation_errors"], rtol=1e-4, atol=1e-5)
    # two LM iterations (accept / reject decisions can only differ on knife-edge trust ratios)
    ref = R.solve(oracle, md, cfg, seeds, gp, gq, idx)
    solver._lm_iteration()
    solver._lm_iteration()
    torch.cuda.synchronize()
    q = solver.q.cpu().numpy()
    close = np.abs(q - ref["joint_position"]).max(-1) < 2e-4
    assert close.mean() > 0.95, close.mean()
    np.testing.assert_allclose(solver.lambda_damping.cpu().numpy()[close], ref["lambda_damping"][close], rtol=1e-5)


@pytest.mark.parametrize("graph,fused", [(False, False), (True, False), (False, True)])
def test_seed_ik_solves_reachable_goals(oracle, device, graph, fused):
    """``fused``: every block of LM iterations as one launch (curobo_hip_seed_ik_iterate, state in LDS) instead of
    five launches per iteration"""
    from oracle import seed_ik_ref as R

    P, S = 40, 16
    model, solver = _solver(device, P, S, use_cuda_graph=graph, batch_success_threshold=2.0, fused_iterations=fused)
    md, gp, gq, seeds, idx = _problem(oracle, model, P, S, seed=5)
    ref = R.solve(oracle, md, R.SeedIKRefCfg(), seeds, gp, gq, idx)
    res = solver.solve_batch(torch.as_tensor(gp[:, :, 0]), torch.as_tensor(gq[:, :, 0]),
                             seed_config=torch.as_tensor(seeds).view(P, S, -1))
    torch.cuda.synchronize()
    assert res.iterations == 16
    ok_ref = ref["final_success"].reshape(P, S).any(-1)
    ok = res.success[:, 0].cpu().numpy()
    assert abs(ok.mean() - ok_ref.mean()) <= 0.05 and ok.mean() > 0.8, (ok.mean(), ok_ref.mean())
    # reported solutions really reach the goals (checked with the oracle's FK) and respect the limits
    sol = res.solution[:, 0].cpu().numpy()
    fk = oracle.kinematics_forward(sol, md, compute_spheres=False)
    err = np.linalg.norm(fk["link_pos"][:, 0] - gp[:, 0, 0], axis=-1)
    assert (err[ok] < 0.005 + 1e-5).all()
    np.testing.assert_allclose(err[ok], res.position_error[:, 0].cpu().numpy()[ok], atol=1e-5)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    assert ((sol[ok] > lo) & (sol[ok] < hi)).all()
    # per-seed agreement with the CPU iteration: most seeds end at the same configuration
    same = np.abs(solver.q.cpu().numpy() - ref["joint_position"]).max(-1) < 1e-3
    assert same.mean() > 0.7, same.mean()


@pytest.mark.parametrize("fused", [False, True])
def test_early_exit_and_sampled_seeds(oracle, device, fused):
    """``fused``: the exit test between blocks runs on the device (curobo_hip_seed_ik_batch_status)"""
    P, S = 16, 32
    model, solver = _solver(device, P, S, fused_iterations=fused)
    md, gp, gq, _, _ = _problem(oracle, model, P, S, seed=9)
    res = solver.solve_batch(torch.as_tensor(gp[:, :, 0]), torch.as_tensor(gq[:, :, 0]), return_seeds=3)
    assert res.solution.shape == (P, 3, 7) and res.success.shape == (P, 3)
    assert res.success[:, 0].float().mean() > 0.9
    assert res.iterations in (4, 8, 12, 16)
    # ranked: successful solutions first, ascending error
    e = (res.position_error + res.rotation_error + 1e10 * (~res.success).float()).cpu().numpy()
    assert (np.diff(e, axis=1) >= 0).all()


def test_update_state_kernel_matches_reference_golden(device):
    """the HIP state-update kernel against the reference's own SeedIterationStateManager outputs
    (tests/golden/seed_ik_update_golden.npz).  The kernel also builds the joint-limit rows; here
    the candidate lies inside the limits' violation-free region or not as drawn, so the golden
    candidate Jacobian / J^T e are fed through weight 0 rows: pose block = the golden candidate."""
    import os

    from curobo_amd.backends import linalg

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seed_ik_update_golden.npz"))
    pick = lambda pre: {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(pre + "/")}  # noqa: E731
    c, cur, cand, out = pick("cfg"), pick("cur"), pick("cand"), pick("out")
    n, D = cand["joint_position"].shape
    R = cand["jacobian"].shape[1]
    T = (R - D) // 6
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dt)  # noqa: E731
    st = {k: t(v) for k, v in cur.items()}
    succ, imp = torch.zeros(n, dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.uint8, device=device)
    # joint_limit_weight = 0: no joint-limit contribution, so error norm / J^T e are the pose block's;
    # the pose cost is fed as one number per problem (its sum is what the kernel uses)
    pose_cost = np.zeros((n, T, 2), np.float32)
    pose_cost[:, 0, 0] = cand["error_norm"]
    pd = np.repeat(cand["position_errors"][:, None], T, 1)
    rd = np.repeat(cand["orientation_errors"][:, None], T, 1)
    linalg.seed_ik_update_state(
        st["joint_position"], st["jacobian"], st["jTerror"], st["error_norm"], st["position_errors"], st["orientation_errors"],
        st["lambda_damping"], succ, imp, t(cand["joint_position"]), t(cand["jacobian"][:, :6 * T]), t(cand["jTerror"]), t(pose_cost),
        t(pd), t(rd), t(g["pred"]), t(g["lo"]), t(g["hi"]), None, None, None, 0.0, float(c["rho_min"]), float(c["lambda_factor"]),
        float(c["lambda_min"]), float(c["lambda_max"]), float(c["convergence_position_tolerance"]),
        float(c["convergence_orientation_tolerance"]), float(c["convergence_joint_limit_weight"]), False)
    torch.cuda.synchronize()
    assert np.array_equal(imp.cpu().numpy().astype(bool), out["improvement"])
    assert np.array_equal(succ.cpu().numpy().astype(bool), out["success"])
    acc = out["improvement"]
    for k in ("joint_position", "jTerror", "position_errors", "orientation_errors", "error_norm"):
        np.testing.assert_array_equal(st[k].cpu().numpy(), out[k], err_msg=k)
    np.testing.assert_allclose(st["lambda_damping"].cpu().numpy(), out["lambda_damping"], rtol=1e-6)
    J = st["jacobian"].cpu().numpy()
    np.testing.assert_array_equal(J[:, :6 * T], out["jacobian"][:, :6 * T])        # pose rows: candidate where accepted, else kept
    np.testing.assert_array_equal(J[~acc, 6 * T:], cur["jacobian"][~acc, 6 * T:])   # rejected: joint-limit rows untouched
    assert (J[acc, 6 * T:] == 0).all()                                               # accepted: rebuilt (weight 0 -> zero rows)


def test_velocity_clamped_bounds_kernel_and_solver(oracle, device):
    """``current_position`` + ``dt``: the joint-limit rows use the bounds one step can reach
    (reference seed_ik_error_calculator.py:355-363).  (1) the state-update kernel's joint-limit block
    against the reference's own outputs (tests/golden/seed_ik_limits_golden.npz); (2) one solver
    evaluation against the oracle; (3) a solve stays within ``velocity_limits * dt`` of the start."""
    import os

    from conftest import GOLDEN_DIR
    from curobo_amd.backends import linalg
    from oracle import seed_ik_ref as R

    g = np.load(os.path.join(GOLDEN_DIR, "seed_ik_limits_golden.npz"))
    n, D = g["q"].shape
    T = 1
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dt)  # noqa: E731
    z = lambda *s: torch.zeros(*s, device=device)  # noqa: E731
    for name, clamp in (("plain", False), ("clamped", True)):
        st = dict(q=z(n, D), J=z(n, 6 * T + D, D), jte=z(n, D), en=z(n), pe=z(n), oe=z(n), lam=torch.ones(n, device=device))
        succ, imp = torch.zeros(n, dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.uint8, device=device)
        extra = (t(g["current_position"]), t(g["dt"]), t(g["velocity_limits"])) if clamp else (None, None, None)
        linalg.seed_ik_update_state(st["q"], st["J"], st["jte"], st["en"], st["pe"], st["oe"], st["lam"], succ, imp, t(g["q"]),
                                    z(n, 6 * T, D), z(n, D), z(n, T, 2), z(n, T), z(n, T), None, t(g["lo"]), t(g["hi"]), *extra,
                                    float(g["weight"]), 1e-3, 2.0, 1e-5, 1e10, 1e-5, 1e-5, 1.0, True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(st["jte"].cpu().numpy(), g[f"{name}/jTerror"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(st["J"].cpu().numpy()[:, 6 * T:], g[f"{name}/jacobian"])
        np.testing.assert_allclose(st["en"].cpu().numpy(), g[f"{name}/error"], rtol=1e-6, atol=1e-7)

    P, S = 10, 8
    model, solver = _solver(device, P, S, use_cuda_graph=True, batch_success_threshold=2.0)
    md = model.as_dict()
    rng = np.random.default_rng(4)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    vmax = np.asarray(md["joint_limits_velocity"], np.float32)[1]
    dt = 0.2
    cur = (lo + (hi - lo) * (0.25 + 0.5 * rng.random((P, 7)))).astype(np.float32)
    target = cur + (0.5 * vmax * dt * (2 * rng.random((P, 7)) - 1)).astype(np.float32)
    fk = oracle.kinematics_forward(target, md, compute_spheres=False)
    gp, gq = fk["link_pos"].reshape(P, 1, 1, 3), fk["link_quat"].reshape(P, 1, 1, 4)
    # (2) one evaluation with clamped bounds == the oracle's
    seeds = (lo + (hi - lo) * rng.random((P * S, 7))).astype(np.float32)
    idx = np.repeat(np.arange(P, dtype=np.int32), S)
    ref0 = R.evaluate(oracle, md, R.SeedIKRefCfg(), seeds, gp, gq, idx, current_position=np.repeat(cur, S, axis=0),
                      dt=np.full(P * S, dt, np.float32))
    plain = R.evaluate(oracle, md, R.SeedIKRefCfg(), seeds, gp, gq, idx)
    assert np.abs(ref0["error_norm"] - plain["error_norm"]).max() > 0.1  # the clamping matters for random seeds
    solver.goal_position.copy_(t(gp))
    solver.goal_quat.copy_(t(gq))
    solver._vel_active = True
    solver._vel_current.copy_(t(np.repeat(cur, S, axis=0)))
    solver._vel_dt.fill_(dt)
    solver._evaluate_candidate(t(seeds), initial=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(solver.error_norm.cpu().numpy(), ref0["error_norm"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(solver.jTerror.cpu().numpy(), ref0["jTerror"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(solver.jacobian.cpu().numpy(), ref0["jacobian"], rtol=1e-4, atol=2e-6)
    # (3) solve from the current position: reaches the goal without leaving the one-step box
    res = solver.solve_batch(t(gp[:, :, 0]), t(gq[:, :, 0]), current_position=t(cur), dt=dt)
    ok = res.success[:, 0].cpu().numpy()
    assert ok.mean() >= 0.9
    step = np.abs(res.solution[:, 0].cpu().numpy() - cur)
    assert (step[ok] <= vmax * dt + 2e-3).all(), step[ok].max(0)
    # without dt the same call is unconstrained (separate captured graph), still solves
    res2 = solver.solve_batch(t(gp[:, :, 0]), t(gq[:, :, 0]), current_position=t(cur))
    assert res2.success[:, 0].float().mean().item() >= 0.9


def test_velocity_and_acceleration_residual_rows(oracle, device):
    """velocity / acceleration regularisation blocks of the seed-IK error (cfg.velocity_weight / acceleration_weight):
    (1) the state-update kernel vs the reference's own _compute_velocity_errors / _compute_acceleration_errors
    (tests/golden/seed_ik_velacc_golden.npz): J^T r, error norm, and the folded diagonal row sqrt(jv^2 + ja^2);
    (2) a solve with the rows on stays closer to the current position than one without."""
    import os

    from curobo_amd.backends import linalg

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seed_ik_velacc_golden.npz"))
    q, cur, vel, dt = g["q"], g["current_position"], g["current_velocity"], g["dt"]
    n, D = q.shape
    T, R = 1, 6 + D
    t = lambda a, dtp=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=device, dtype=dtp)  # noqa: E731
    z = lambda *s: torch.zeros(*s, device=device)  # noqa: E731
    for wv, wa in ((float(g["velocity_weight"]), 0.0), (0.0, float(g["acceleration_weight"])),
                   (float(g["velocity_weight"]), float(g["acceleration_weight"]))):
        st_q, J, jTe, en, pe, oe, lam = z(n, D), z(n, R, D), z(n, D), z(n), z(n), z(n), torch.full((n,), 0.2, device=device)
        succ, imp = torch.zeros(n, dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.uint8, device=device)
        big = np.full(D, 1e6, np.float32)
        linalg.seed_ik_update_state(st_q, J, jTe, en, pe, oe, lam, succ, imp, t(q), z(n, 6, D), z(n, D), z(n, T, 2), z(n, T), z(n, T), None,
                                    t(-big), t(big), t(cur), t(dt), None, 1.0, 1e-3, 2.0, 1e-5, 1e10, 1e-5, 1e-5, 0.0, True,
                                    current_velocity=t(vel), velocity_weight=wv, acceleration_weight=wa)
        torch.cuda.synchronize()
        want_jt = (g["vel_jTerror"] if wv > 0 else 0) + (g["acc_jTerror"] if wa > 0 else 0)
        want_en = (g["vel_error"] if wv > 0 else 0) + (g["acc_error"] if wa > 0 else 0)
        want_d2 = (g["vel_jacobian_diag"] ** 2 if wv > 0 else 0) + (g["acc_jacobian_diag"] ** 2 if wa > 0 else 0)
        np.testing.assert_allclose(jTe.cpu().numpy(), want_jt, rtol=2e-5, atol=1e-5 * np.abs(want_jt).max())
        np.testing.assert_allclose(en.cpu().numpy(), want_en, rtol=2e-5)
        Jn = J.cpu().numpy()
        diag = Jn[:, 6 + np.arange(D), np.arange(D)]
        np.testing.assert_allclose(diag ** 2, want_d2, rtol=5e-5)
        off = Jn[:, 6:].copy()
        off[:, np.arange(D), np.arange(D)] = 0
        assert (off == 0).all() and (Jn[:, :6] == 0).all()
    # ---- (2) the rows pull the solution towards the current state
    P, S = 12, 8
    md, gp, gq, _, _ = _problem(oracle, load_model("franka"), P, S, seed=4)
    rng = np.random.default_rng(5)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    qc = torch.as_tensor((0.5 * (lo + hi) + 0.2 * (hi - lo) * rng.uniform(-1, 1, size=(P, lo.shape[0]))).astype(np.float32), device=device)
    dist = {}
    for w in (0.0, 0.5):
        _, sol = _solver(device, P, S, velocity_weight=w, acceleration_weight=0.01 * w, batch_success_threshold=2.0)
        r = sol.solve_batch(torch.as_tensor(gp, device=device), torch.as_tensor(gq, device=device), current_position=qc, dt=10.0,
                            current_velocity=torch.zeros_like(qc))
        # a soft regulariser trades pose accuracy for staying close: best seed by pose error, whatever the tolerance says
        dist[w] = float((r.solution[:, 0] - qc).norm(dim=-1).mean())
        if w == 0.0:
            assert float(r.success[:, 0].float().mean()) >= 0.6
    assert dist[0.5] < 0.9 * dist[0.0], dist


def test_seed_selection_and_ik_ranking_kernels(device):
    """curobo_hip_seed_ik_select / curobo_hip_ik_rank against the torch formulation of the reference's selection
    (seed_ik_solver.py:522-572, solver_ik.py:440-580) on random data with ties: exact indices (ties -> lower index)"""
    from curobo_amd.backends import linalg

    rng = np.random.default_rng(5)
    P, S, D, T, k = 37, 96, 7, 2, 11
    lo, hi = -np.ones(D, np.float32), np.ones(D, np.float32)
    q = rng.uniform(-1.1, 1.1, size=(P, S, D)).astype(np.float32)
    pos = rng.choice([0.001, 0.002, 0.004, 0.02], size=(P, S)).astype(np.float32)  # few distinct values: many ties
    ori = rng.choice([0.01, 0.03, 0.2], size=(P, S)).astype(np.float32)
    cur = rng.uniform(-1, 1, size=(P, D)).astype(np.float32)
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    for use_cur in (False, True):
        ok = (pos < 0.005) & (ori < 0.05) & ((q > lo) & (q < hi)).all(-1)
        cost = pos + ori
        if use_cur:
            cost = cost + np.float32(0.01) * np.linalg.norm(q - cur[:, None], axis=-1).astype(np.float32)
        cost = cost + np.float32(1e10) * (~ok).astype(np.float32)
        order = np.argsort(cost, axis=1, kind="stable")[:, :k]
        o_ok = torch.zeros(P, k, dtype=torch.uint8, device=device)
        o_sol, o_pos, o_ori = torch.zeros(P, k, D, device=device), torch.zeros(P, k, device=device), torch.zeros(P, k, device=device)
        linalg.seed_ik_select(o_ok, o_sol, o_pos, o_ori, t(q), t(pos), t(ori), t(lo), t(hi), t(cur) if use_cur else None,
                              0.005, 0.05, 0.01, True, k)
        torch.cuda.synchronize()
        ar = np.arange(P)[:, None]
        if not use_cur:  # (with the distance term the costs are distinct up to rounding of the norm: compare values)
            np.testing.assert_array_equal(o_sol.cpu().numpy(), q[ar, order])
        np.testing.assert_array_equal(o_ok.cpu().numpy().astype(bool), ok[ar, order])
        np.testing.assert_array_equal(o_pos.cpu().numpy(), pos[ar, order])
    # ---- IK ranking
    n_sc = 9
    cst = rng.choice([1.0, 2.0, 3.5], size=(P, S)).astype(np.float32)
    # (per tool frame: a solution converges only when EVERY frame does, the reported error is the largest over the frames)
    pd = np.repeat(pos[..., None], T, -1).copy()
    rd = np.repeat(ori[..., None], T, -1).copy()
    if T > 1:
        pd[..., 1:] = rng.choice([0.001, 0.004, 0.02], p=[0.6, 0.3, 0.1], size=pd[..., 1:].shape).astype(np.float32)
        rd[..., 1:] = rng.choice([0.01, 0.03, 0.2], p=[0.6, 0.3, 0.1], size=rd[..., 1:].shape).astype(np.float32)
    selfd = (rng.random((P, S)) < 0.1).astype(np.float32) * 0.3
    csp = (rng.random((P, S, D)) < 0.02).astype(np.float32)
    scd = (rng.random((P, S, n_sc)) < 0.01).astype(np.float32) * 0.1
    gidx = rng.integers(0, 3, size=(P, S, T)).astype(np.int32)
    ok = (selfd <= 0) & (csp.sum(-1) <= 0) & (scd.sum(-1) <= 0) & (pd < 0.005).all(-1) & (rd < 0.05).all(-1)
    ranked = cst + np.float32(1e16) * (~ok).astype(np.float32)
    order = np.argsort(ranked, axis=1, kind="stable")[:, :k]
    o_ok = torch.zeros(P, k, dtype=torch.uint8, device=device)
    o_sol = torch.zeros(P, k, D, device=device)
    o_pe, o_re, o_c = (torch.zeros(P, k, device=device) for _ in range(3))
    o_si, o_gi = torch.zeros(P, k, dtype=torch.int64, device=device), torch.zeros(P, k, T, dtype=torch.int64, device=device)
    linalg.ik_rank(o_ok, o_sol, o_pe, o_re, o_c, o_si, o_gi, t(q).view(P * S, D), t(cst).view(-1), t(pd).view(P * S, T),
                   t(rd).view(P * S, T), t(selfd).view(-1), t(csp).view(P * S, D), t(scd), t(gidx).view(P * S, T), 0.005, 0.05,
                   P, S, k, 1000)
    torch.cuda.synchronize()
    ar = np.arange(P)[:, None]
    np.testing.assert_array_equal(o_si.cpu().numpy(), order + 1000)
    np.testing.assert_array_equal(o_ok.cpu().numpy().astype(bool), ok[ar, order])
    np.testing.assert_array_equal(o_sol.cpu().numpy(), q[ar, order])
    np.testing.assert_array_equal(o_c.cpu().numpy(), cst[ar, order])
    np.testing.assert_array_equal(o_gi.cpu().numpy(), gidx[ar, order])  # one member index per tool frame
    np.testing.assert_array_equal(o_pe.cpu().numpy(), pd.max(-1)[ar, order])
    np.testing.assert_array_equal(o_re.cpu().numpy(), rd.max(-1)[ar, order])
    assert ok.any(1).mean() > 0.5 and (~ok).any()


def test_argmin_rows_kernel_matches_torch(device):
    """curobo_hip_argmin_rows (the local stage of the arg-min exchange) == torch.min + gather, ties -> first index"""
    from curobo_amd.distributed import global_argmin

    rng = np.random.default_rng(2)
    P, S, V = 19, 300, 84
    cost = rng.choice([0.5, 1.0, 2.0, 7.0], size=(P, S)).astype(np.float32)  # many ties
    cost[3] = np.nan
    cost[4, :10] = np.nan
    pay = rng.normal(size=(P, S, V)).astype(np.float32)
    c, i, x = global_argmin(torch.as_tensor(cost, device=device), torch.as_tensor(pay, device=device), 1000)
    torch.cuda.synchronize()
    ref_i = np.array([int(np.nanargmin(r)) if not np.isnan(r).all() else 0 for r in cost])
    np.testing.assert_array_equal(i.cpu().numpy(), ref_i + 1000)
    np.testing.assert_array_equal(x.cpu().numpy(), pay[np.arange(P), ref_i])
    np.testing.assert_array_equal(c.cpu().numpy()[np.arange(P) != 3], cost[np.arange(P), ref_i][np.arange(P) != 3])
