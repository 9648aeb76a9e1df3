"""MPPI distribution update kernel vs the reference's torch functions (golden) and the MPPI loop."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "mppi_golden.npz")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_mppi_update_kernel_matches_reference_torch(case, device):
    from curobo_amd.backends import optimization as Op
    from oracle.mppi_ref import mean_cov_diag_a

    g = np.load(GOLD)
    k = lambda n: g[f"c{case}/{n}"]  # noqa: E731
    sm, sc, kappa, beta = [float(x) for x in k("params")]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    costs, actions, mean, cov = t(k("costs")), t(k("actions")), t(k("mean")), t(k("cov"))
    b, p, ha, d = actions.shape
    new_mean, new_cov, new_tril = torch.zeros_like(mean), torch.zeros_like(cov), torch.zeros_like(cov)
    best, w = torch.zeros_like(mean), torch.zeros(b, p, device=device)
    Op.mppi_update_distribution(new_mean, new_cov, new_tril, best, w, costs, t(k("gamma_seq").reshape(-1)), actions,
                                mean, cov, beta, sm, sc, kappa)
    torch.cuda.synchronize()
    np.testing.assert_allclose(w.cpu().numpy(), k("w"), rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(new_mean.cpu().numpy(), k("new_mean"), rtol=1e-4, atol=2e-5)  # sharp softmax: fp32 exp
    np.testing.assert_allclose(new_cov.cpu().numpy(), k("new_cov"), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(new_tril.cpu().numpy(), k("new_tril"), rtol=1e-4, atol=2e-5)
    ref_best = mean_cov_diag_a(k("costs"), k("actions"), k("gamma_seq"), k("mean"), k("cov"), sm, sc, kappa, beta)[4]
    np.testing.assert_array_equal(best.cpu().numpy(), ref_best)


@pytest.mark.parametrize("noise,bar", [("torch", 0.4), ("sample_lib", 1.2)])
def test_mppi_minimises_a_quadratic_and_a_collision_rollout(noise, bar, device):
    from curobo_amd.optim import MPPI, MPPICfg

    torch.manual_seed(0)
    B, Ha, D = 3, 8, 5
    target = torch.linspace(-0.5, 0.5, Ha * D, device=device).view(1, Ha * D)
    lo, hi = -torch.ones(D, device=device), torch.ones(D, device=device)
    cfg = MPPICfg(num_problems=B, num_particles=512, num_iters=40, beta=1.0, init_cov=0.3, noise=noise)
    opt = MPPI(cfg, lambda a: ((a - target) ** 2).sum(-1), Ha, D, (lo, hi), device)
    out = opt.optimize(torch.zeros(B, Ha, D, device=device))
    torch.cuda.synchronize()
    # start: |target|^2 = 3.5; the softmax-weighted mean walks to the optimum (40-dim problem, beta 1).  Fresh white noise every
    # iteration gets below 0.4; the reference's sample library -- ONE set of 512 Halton particles for all iterations, smoothed
    # in time by its three-tap filter -- explores this synthetic ramp target more slowly (measured 0.95)
    assert float(((out.view(B, -1) - target) ** 2).sum(-1).max()) < bar
    assert float(opt.cov.max()) < 0.3 + 1e-6
    # deterministic given the seed
    opt2 = MPPI(cfg, lambda a: ((a - target) ** 2).sum(-1), Ha, D, (lo, hi), device)
    assert torch.equal(out, opt2.optimize(torch.zeros(B, Ha, D, device=device)))


def test_mppi_on_collision_rollout_reduces_cost(device):
    """the MPPI step on the path's own rollout: cost-only evaluation of B-spline knot particles"""
    from conftest import load_model
    from curobo_amd.optim import MPPI, MPPICfg
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    rcfg = CollisionRolloutCfg()
    B, P = 4, 128
    ro = CollisionRollout(kin, scene, B * P, rcfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    knots = torch.as_tensor(seed_knots(model, B, rcfg.n_knots, seed=3), device=device)
    cost_fn = lambda a: ro.cost_and_gradient(a)[0].clone()  # noqa: E731
    c0 = cost_fn(knots.view(B, 1, -1).expand(B, P, -1).reshape(B * P, -1).contiguous()).view(B, P)[:, 0]
    cfg = MPPICfg(num_problems=B, num_particles=P, num_iters=25, beta=1000.0, init_cov=0.02, sample_mode="BEST")
    opt = MPPI(cfg, cost_fn, rcfg.n_knots, kin.num_dof, (kin.joint_limits_position[0], kin.joint_limits_position[1]), device)
    best = opt.optimize(knots)
    c1 = cost_fn(best.view(B, 1, -1).expand(B, P, -1).reshape(B * P, -1).contiguous()).view(B, P)[:, 0]
    torch.cuda.synchronize()
    assert (c0 > 0).any()
    assert (c1 <= c0 + 1e-3).all() and float(c1.sum()) < 0.7 * float(c0.sum())


def test_multi_stage_mppi_then_lbfgs(device):
    """reference MultiStageOptimizer: a particle stage seeds the gradient stage; the chain ends at
    least as low as either stage alone started, and a disabled stage is skipped."""
    from curobo_amd.optim import MPPI, LBFGSOpt, LBFGSOptCfg, MPPICfg, MultiStageOptimizer

    B, Ha, D = 4, 6, 3
    V = Ha * D
    target = torch.linspace(-0.6, 0.6, V, device=device).view(1, V)
    lo, hi = -torch.ones(D, device=device), torch.ones(D, device=device)
    cost = lambda a: ((a - target) ** 2).sum(-1)  # noqa: E731
    mppi = MPPI(MPPICfg(num_problems=B, num_particles=256, num_iters=10, beta=1.0, init_cov=0.3), cost, Ha, D, (lo, hi), device)
    ocfg = LBFGSOptCfg(num_problems=B, num_iters=20, inner_iters=10, history=5)
    nls = len(ocfg.line_search_scale)

    def cost_grad(x):
        return cost(x).clone(), (2.0 * (x - target)).contiguous()
    lb = LBFGSOpt(ocfg, cost_grad, Ha, D, (lo, hi), device, use_cuda_graph=False)
    assert nls == 4
    chain = MultiStageOptimizer([mppi, lb])
    assert chain.solver_names == ["MPPI", "LBFGSOpt"] and chain.action_horizon == Ha and chain.opt_dim == V
    x0 = torch.zeros(B, Ha, D, device=device)
    after_mppi = mppi.optimize(x0).clone()
    mppi.reset_distribution()
    out = chain.optimize(x0)
    torch.cuda.synchronize()
    c_m, c_out = cost(after_mppi.view(B, V)), cost(out.view(B, V))
    assert float(c_out.max()) < 1e-3 and bool((c_out <= c_m + 1e-6).all())
    chain.enable_stage(0, False)  # gradient stage alone from the same seed: also converges here
    assert float(cost(chain.optimize(x0).view(B, V)).max()) < 1e-3
