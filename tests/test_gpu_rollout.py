"""End-to-end parity of the rollout hot path and the L-BFGS loop on the GPU vs the oracle."""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def _setup(device, seeds=24):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(arrays, device)
    cfg = CollisionRolloutCfg()
    knots = seed_knots(model, seeds, cfg.n_knots, seed=7)
    start = start_configuration(model)
    ro = CollisionRollout(kin, scene, seeds, cfg)
    ro.update_start_state(torch.as_tensor(start, device=device))
    return model, kin, arrays, cfg, knots, start, ro


def test_rollout_cost_and_gradient_matches_oracle(oracle, device):
    from oracle.rollout_ref import rollout_cost_and_gradient

    model, kin, arrays, cfg, knots, start, ro = _setup(device)
    ref = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start)
    x = torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1)
    cost, grad = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    assert (ref["cost"] > 0).mean() > 0.5, "workload must be in collision for a meaningful check"
    np.testing.assert_allclose(ro.position.cpu().numpy(), ref["position"], atol=1e-5)
    np.testing.assert_allclose(ro.robot_spheres.cpu().numpy(), ref["robot_spheres"], atol=1e-5)
    np.testing.assert_allclose(ro.self_dist.cpu().numpy()[..., 0], ref["self_cost"], atol=1e-5, rtol=1e-5)
    # costs carry the reference weights (1e5 / 1e4): compare relative to their scale
    sc = ref["scene_cost"]
    np.testing.assert_allclose(ro.scene_dist.cpu().numpy(), sc, atol=1e-5 * max(1.0, sc.max()), rtol=1e-4)
    np.testing.assert_allclose(cost.cpu().numpy(), ref["cost"], rtol=1e-4, atol=1e-3)
    gq = ref["grad_q"]
    np.testing.assert_allclose(ro.grad_q.cpu().numpy(), gq, atol=2e-5 * np.abs(gq).max(), rtol=2e-3)
    gk = ref["grad_knots"]
    np.testing.assert_allclose(grad.cpu().numpy().reshape(gk.shape), gk, atol=2e-5 * np.abs(gk).max(), rtol=2e-3)


def test_rollout_second_call_is_identical(device):
    """stateful buffers (self-collision sparse flags, scene buffers) must not leak between calls"""
    model, kin, arrays, cfg, knots, start, ro = _setup(device)
    x = torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1)
    c1, g1 = [t.clone() for t in ro.cost_and_gradient(x)]
    ro.cost_and_gradient(x * 0.5)
    c2, g2 = ro.cost_and_gradient(x)
    assert torch.equal(c1, c2) and torch.equal(g1, g2)


def test_lbfgs_reduces_cost_and_graph_matches_eager(device):
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
    from curobo_amd.rollout import CollisionRollout

    seeds = 16
    model, kin, arrays, cfg, knots, start, _ = _setup(device, seeds)
    results = []
    for use_graph in (False, True):
        ocfg = LBFGSOptCfg(num_problems=seeds, inner_iters=5, num_iters=20)
        ro = CollisionRollout(kin, _scene(arrays, device), seeds * 4, cfg)
        ro.update_start_state(torch.as_tensor(start, device=device))
        opt = LBFGSOpt(ocfg, ro.cost_and_gradient, cfg.n_knots, kin.num_dof,
                       (kin.joint_limits_position[0], kin.joint_limits_position[1]), device, use_cuda_graph=use_graph)
        seed = torch.as_tensor(knots, device=device)
        opt.reinitialize(seed)
        c0 = opt.best_cost.clone()
        best = opt.optimize(seed)
        torch.cuda.synchronize()
        assert torch.isfinite(best).all()
        assert (opt.best_cost <= c0 + 1e-6).all(), "best cost must never increase"
        assert (opt.best_cost < c0).float().mean() > 0.5, "most seeds should improve"
        assert int(opt.current_iteration[0]) == 20
        results.append((opt.best_cost.clone(), best.clone()))
    assert torch.equal(results[0][0], results[1][0]), "hipGraph replay must reproduce the eager run bit-for-bit"
    assert torch.equal(results[0][1], results[1][1])


def _scene(arrays, device):
    from curobo_amd.scene import SceneData

    return SceneData.from_arrays(arrays, device)


def test_global_argmin_single_process(device):
    from curobo_amd.distributed import global_argmin

    cost = torch.tensor([[3.0, 1.0, 1.0, 5.0], [0.5, 2.0, 0.25, 0.25]], device=device)
    payload = torch.arange(2 * 4 * 3, device=device, dtype=torch.float32).view(2, 4, 3)
    c, i, p = global_argmin(cost, payload, seed_offset=10)
    assert c.tolist() == [1.0, 0.25] and i.tolist() == [11, 12]
    assert torch.equal(p, payload[[0, 1], [1, 2]])
