"""CPU: the seed-IK restatement converges on reachable goals, and the Halton seed buffer is the
reference's (scipy scrambled Halton, same seed -> same points; checked against the reference's own
HaltonSequencer when the checkout is present)."""

import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_model


def test_seed_ik_restatement_converges(oracle):
    from oracle import seed_ik_ref as R

    md = load_model("franka").as_dict()
    rng = np.random.default_rng(0)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    P, S = 20, 8
    qg = (lo + (hi - lo) * rng.random((P, 7))).astype(np.float32)
    fk = oracle.kinematics_forward(qg, md, compute_spheres=False)
    seeds = (lo + (hi - lo) * rng.random((P * S, 7))).astype(np.float32)
    idx = np.repeat(np.arange(P, dtype=np.int32), S)
    st = R.solve(oracle, md, R.SeedIKRefCfg(), seeds, fk["link_pos"].reshape(P, 1, 1, 3), fk["link_quat"].reshape(P, 1, 1, 4), idx)
    ok = st["final_success"].reshape(P, S)
    assert ok.any(1).mean() >= 0.85
    # what is flagged solved is solved: FK of the solution is at the goal
    sol = st["joint_position"][st["final_success"]]
    goal = np.repeat(fk["link_pos"][:, 0], S, axis=0)[st["final_success"]]
    err = np.linalg.norm(oracle.kinematics_forward(sol, md, compute_spheres=False)["link_pos"][:, 0] - goal, axis=-1)
    assert (err < 0.005 + 1e-6).all()
    # the trust-region logic moved lambda both ways
    assert st["lambda_damping"].min() < 0.2 < st["lambda_damping"].max()


def test_joint_limit_block_matches_reference_golden():
    """oracle.seed_ik_ref.joint_limit_block == the reference's own _compute_joint_limit_errors
    (tests/golden/make_seed_ik_limits_golden.py), plain and with velocity-clamped bounds"""
    from conftest import GOLDEN_DIR
    from oracle import seed_ik_ref as R

    g = np.load(os.path.join(GOLDEN_DIR, "seed_ik_limits_golden.npz"))
    for name, kw in (("plain", {}), ("clamped", dict(current_position=g["current_position"], dt=g["dt"],
                                                      velocity_limits=g["velocity_limits"]))):
        jte, diag, err = R.joint_limit_block(g["q"], g["lo"], g["hi"], float(g["weight"]), **kw)
        np.testing.assert_array_equal(jte, g[f"{name}/jTerror"])
        np.testing.assert_array_equal(diag, np.diagonal(g[f"{name}/jacobian"], axis1=-2, axis2=-1))
        np.testing.assert_allclose(err, g[f"{name}/error"], rtol=1e-6)
    assert (g["clamped/jTerror"] != g["plain/jTerror"]).any()


def test_halton_seed_buffer_is_scipys_scrambled_halton():
    from scipy.stats.qmc import Halton

    from curobo_amd.solver.seed_ik import HaltonSeeds

    lo, hi = torch.tensor([-1.0, 0.0, 2.0]), torch.tensor([1.0, 0.5, 4.0])
    s = HaltonSeeds(3, lo, hi, seed=451)
    np.testing.assert_allclose(s.buffer.numpy(), Halton(d=3, seed=451, scramble=True).random(2000).astype(np.float32))
    a = s.get_samples(64)
    assert a.shape == (64, 3) and bool(((a >= lo) & (a <= hi)).all())
    s.reset()
    assert torch.equal(a, s.get_samples(64))
    ref_root = "/root/reference"
    if os.path.isdir(ref_root):
        sys.path.insert(0, ref_root)
        try:
            from curobo._src.util.sampling.sequencer_halton import HaltonSequencer
        except Exception as e:  # optional dependency of the reference missing
            pytest.skip(f"reference sampler not importable: {e}")
        finally:
            sys.path.remove(ref_root)
        np.testing.assert_allclose(s.buffer.numpy(), HaltonSequencer(ndims=3, seed=451).random(2000).astype(np.float32))


def _golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seed_ik_update_golden.npz"))
    pick = lambda pre: {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(pre + "/")}  # noqa: E731
    return g, pick("cfg"), pick("cur"), pick("cand"), pick("out")


def test_state_update_restatement_matches_reference_golden():
    """oracle/seed_ik_ref.update_state against the outputs of the reference's own
    SeedIterationStateManager.update_iteration_state (tests/golden/make_seed_ik_golden.py)"""
    from oracle import seed_ik_ref as R

    g, c, cur, cand, out = _golden()
    cfg = R.SeedIKRefCfg(rho_min=float(c["rho_min"]), lambda_factor=float(c["lambda_factor"]), lambda_min=float(c["lambda_min"]),
                         lambda_max=float(c["lambda_max"]), convergence_position_tolerance=float(c["convergence_position_tolerance"]),
                         convergence_orientation_tolerance=float(c["convergence_orientation_tolerance"]),
                         convergence_joint_limit_weight=float(c["convergence_joint_limit_weight"]))
    got = R.update_state(cur, cand, g["pred"], g["lo"], g["hi"], cfg)
    assert np.array_equal(got["improvement"], out["improvement"]) and np.array_equal(got["success"], out["success"])
    for k in ("joint_position", "jacobian", "jTerror", "error_norm", "position_errors", "orientation_errors"):
        np.testing.assert_array_equal(got[k], out[k], err_msg=k)
    np.testing.assert_allclose(got["lambda_damping"], out["lambda_damping"], rtol=1e-6)
    assert 0 < out["improvement"].sum() < len(out["improvement"]) and out["success"].any()
