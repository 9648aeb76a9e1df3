"""End-to-end parity of the rollout hot path and the L-BFGS loop on the GPU vs the oracle."""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def _setup(device, seeds=24, fused=False):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(arrays, device)
    cfg = CollisionRolloutCfg(use_fused=fused)
    knots = seed_knots(model, seeds, cfg.n_knots, seed=7)
    start = start_configuration(model)
    ro = CollisionRollout(kin, scene, seeds, cfg)
    ro.update_start_state(torch.as_tensor(start, device=device))
    return model, kin, arrays, cfg, knots, start, ro


def test_rollout_cost_and_gradient_matches_oracle(oracle, device):
    """Stage-by-stage parity: every kernel of the rollout is compared with the oracle applied to
    the GPU's own upstream output (identical inputs -> 1e-5-class agreement), then the composed
    result is compared end to end.  The swept scene cost is discontinuous in its inputs (the
    adaptive sweep `break`s on float comparisons, reference wp_sweep_collision_kernel.py:188-209),
    so end-to-end -- where FK rounding differs in the last bit -- a handful of samples may take a
    different branch; that comparison therefore allows 1% outlier trajectories."""
    from oracle.rollout_ref import rollout_cost_and_gradient

    model, kin, arrays, cfg, knots, start, ro = _setup(device)
    md = model.as_dict()
    b, nk, d = knots.shape
    ph, S = cfg.padded_horizon, model.num_spheres
    x = torch.as_tensor(knots, device=device).reshape(b, -1)
    cost, grad = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = lambda t: t.cpu().numpy()  # noqa: E731
    ref = rollout_cost_and_gradient(oracle, md, arrays, knots, start)
    assert (ref["cost"] > 0).mean() > 0.5, "workload must be in collision for a meaningful check"
    # 1. transition
    np.testing.assert_allclose(g(ro.position), ref["position"], atol=1e-5)
    # 2. FK from the GPU's positions
    fk = oracle.kinematics_forward(g(ro.position).reshape(b * ph, d), md, horizon=ph)
    np.testing.assert_allclose(g(ro.robot_spheres).reshape(b * ph, S, 4), fk["robot_spheres"], atol=1e-5)
    np.testing.assert_allclose(g(ro.cumul_mat).reshape(fk["cumul_mat"].shape), fk["cumul_mat"], atol=1e-5)
    # 3. self collision on the GPU's spheres (indices exact, distances 1e-5 relative to weight)
    sc = oracle.self_collision(g(ro.robot_spheres), model.sphere_padding, model.collision_pairs,
                               cfg.self_collision_weight)
    assert np.array_equal(g(ro.self_sparse).reshape(b * ph, S), sc["sparse_index"])
    np.testing.assert_allclose(g(ro.self_dist).reshape(-1), sc["distance"], rtol=1e-5, atol=1e-5 * cfg.self_collision_weight * 1e-3)
    np.testing.assert_allclose(g(ro.self_grad).reshape(b * ph, S, 4), sc["gradient"], rtol=1e-5, atol=1e-3)
    # 4. swept scene collision + speed metric on the GPU's spheres
    wc = oracle.scene_collision(g(ro.robot_spheres), arrays, cfg.scene_collision_weight, cfg.activation_distance,
                                sweep=True, enable_speed_metric=True, speed_dt=cfg.traj_dt)
    # The sweep is discontinuous at zero motion: `if jump >= half_dist: break` with jump = 0 adds
    # a duplicate of the centre sample iff half_dist > 0 (wp_sweep_collision_kernel.py:186-189).
    # The first points of a spline that starts at rest are equal up to rounding, so there the
    # sample count (1x, 2x or 3x the centre cost) is decided by the last bit; those spheres are
    # only checked to be within that factor, every moving sphere is checked tightly.
    sp = g(ro.robot_spheres)[..., :3]
    step = np.linalg.norm(np.diff(sp, axis=1), axis=-1)  # [b, ph-1, S]
    still = np.zeros(sp.shape[:3], bool)
    still[:, 1:] |= step < 1e-5
    still[:, :-1] |= step < 1e-5
    mv = ~still
    assert (wc["distance"][mv] > 0).sum() > 100, "moving spheres must include collisions"
    dmax = max(1.0, wc["distance"].max())
    np.testing.assert_allclose(g(ro.scene_dist)[mv], wc["distance"][mv], rtol=2e-4, atol=2e-5 * dmax)
    gmax = max(1.0, np.abs(wc["gradient"]).max())
    np.testing.assert_allclose(g(ro.scene_grad)[mv], wc["gradient"][mv], rtol=2e-3, atol=2e-5 * gmax)
    a_, r_ = g(ro.scene_dist)[still], wc["distance"][still]
    assert np.all((a_ <= 3.001 * r_ + 1e-3) & (r_ <= 3.001 * a_ + 1e-3))
    # 5. per-trajectory sum of the GPU's own cost buffers
    np.testing.assert_allclose(g(cost), oracle.trajectory_cost_sum(g(ro.self_dist)[..., 0], g(ro.scene_dist)), rtol=1e-5)
    # 6. FK backward from the GPU's cumulative transforms and gradient buffers
    gs = g(ro.self_grad) + g(ro.scene_grad) * np.array([1, 1, 1, 0], np.float32)
    gq = oracle.kinematics_backward(md, g(ro.cumul_mat), gs.reshape(b * ph, S, 4), horizon=ph)
    np.testing.assert_allclose(g(ro.grad_q).reshape(gq.shape), gq, rtol=2e-4, atol=2e-6 * np.abs(gq).max())
    # 7. B-spline backward from the GPU's grad_q
    z = np.zeros((b, ph, d), np.float32)
    gk = oracle.bspline_backward(g(ro.grad_q), z, z, z, np.array([cfg.traj_dt], np.float32), np.zeros(b, np.int32),
                                 np.zeros(1, np.uint8), nk, cfg.bspline_degree)
    np.testing.assert_allclose(g(grad).reshape(gk.shape), gk, rtol=2e-4, atol=2e-6 * np.abs(gk).max())
    # end to end against the all-oracle pipeline: robust statistics only, because of the
    # zero-motion discontinuity above (the strict end-to-end check is the non-swept test below)
    rel = np.abs(g(cost) - ref["cost"]) / np.maximum(np.abs(ref["cost"]), 1.0)
    assert np.median(rel) < 1e-4, f"end-to-end cost mismatch: {np.sort(rel)[-5:]}"


def test_rollout_end_to_end_discrete_collision(oracle, device):
    """Without the sweep the whole path is continuous in its inputs, so the composed GPU result is
    compared strictly with the all-oracle pipeline (cost 1e-4 relative, gradients 2e-3)."""
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from oracle.rollout_ref import rollout_cost_and_gradient

    model, kin, arrays, _, knots, start, ro0 = _setup(device)
    cfg = CollisionRolloutCfg(use_sweep=False, use_speed_metric=False)
    ro = CollisionRollout(kin, ro0.scene, knots.shape[0], cfg)
    ro.update_start_state(torch.as_tensor(start, device=device))
    ref = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start, use_sweep=False,
                                    use_speed_metric=False)
    cost, grad = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1))
    torch.cuda.synchronize()
    assert (ref["cost"] > 0).mean() > 0.5
    np.testing.assert_allclose(cost.cpu().numpy(), ref["cost"], rtol=1e-4, atol=1e-2)
    gk = ref["grad_knots"].reshape(knots.shape[0], -1)
    np.testing.assert_allclose(grad.cpu().numpy(), gk, rtol=2e-3, atol=2e-5 * np.abs(gk).max())


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


@pytest.mark.parametrize("fused_rollout", [False, True])
def test_fused_iteration_tail_equals_three_launches(device, fused_rollout):
    """line search + two-loop + next candidates in one launch reproduces the three drop-in launches
    bit for bit (same device functions, operands handed over in registers instead of HBM)"""
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
    from curobo_amd.rollout import CollisionRollout

    seeds = 16
    model, kin, arrays, cfg, knots, start, _ = _setup(device, seeds, fused=fused_rollout)
    state = []
    # (fused tail, overlapped): the workgroup-per-problem form and the wavefront-per-problem form the stream shards use
    for fused_tail, overlapped in ((False, False), (True, False), (True, True)):
        ocfg = LBFGSOptCfg(num_problems=seeds, inner_iters=4, num_iters=12, fused_tail=fused_tail)
        ro = CollisionRollout(kin, _scene(arrays, device), seeds * 4, cfg)
        ro.update_start_state(torch.as_tensor(start, device=device))
        opt = LBFGSOpt(ocfg, ro.cost_and_gradient, cfg.n_knots, kin.num_dof,
                       (kin.joint_limits_position[0], kin.joint_limits_position[1]), device, use_cuda_graph=False)
        opt.overlapped = overlapped
        best = opt.optimize(torch.as_tensor(knots, device=device))
        torch.cuda.synchronize()
        state.append([t.clone() for t in (best, opt.best_cost, opt.exploration_action, opt.exploration_gradient,
                                          opt.step_direction, opt.rho, opt.y, opt.s, opt.best_iteration)])
    for other in state[1:]:
        for a_, b_ in zip(state[0], other):
            assert torch.equal(a_, b_)


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


@pytest.mark.parametrize("V,hist", [(7, 7), (12, 15), (16, 16)])
def test_fused_iteration_tail_row16_matches_three_launches(device, V, hist):
    """IK-sized problems put one problem on one 16-lane row (four per wavefront).  The row
    reductions are the first four DPP steps of the wavefront ladder (the other lanes only add
    zeros), but the two instantiations are compiled separately and FMA contraction differs in the
    last bit, so the comparison is at fp32 rounding level over a few iterations (L-BFGS amplifies
    rounding over long runs); index outputs must agree exactly."""
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg

    torch.manual_seed(0)
    B, N = 37, 4
    A = torch.randn(B, V, V, device=device) * 0.5 + torch.eye(V, device=device)
    bvec = torch.randn(B, V, device=device)
    cost_buf, grad_buf = torch.zeros(B * N, device=device), torch.zeros(B * N, V, device=device)

    def cost_and_gradient(x):  # x [B*N, V]; fixed output buffers like a rollout
        xr = x.view(B, N, V)
        r = (A.unsqueeze(1) * xr.unsqueeze(2)).sum(-1) - bvec.unsqueeze(1)
        cost_buf.copy_((0.5 * (r ** 2).sum(-1)).view(-1))
        grad_buf.copy_((A.unsqueeze(1) * r.unsqueeze(-1)).sum(2).reshape(B * N, V))
        return cost_buf, grad_buf

    state = []
    for fused_tail in (False, True):
        cfg = LBFGSOptCfg(num_problems=B, history=hist, inner_iters=1, num_iters=4, fused_tail=fused_tail)
        opt = LBFGSOpt(cfg, cost_and_gradient, 1, V, (-10 * torch.ones(V, device=device), 10 * torch.ones(V, device=device)),
                       device, use_cuda_graph=False)
        c0 = None
        best = opt.optimize(torch.randn(B, 1, V, device=device, generator=torch.Generator(device=device).manual_seed(1)))
        torch.cuda.synchronize()
        state.append([t.clone() for t in (best, opt.best_cost, opt.exploration_action, opt.step_direction, opt.rho,
                                          opt.y, opt.s, opt.best_iteration, opt.selected_idx)])
    for a_, b_ in zip(state[0][:7], state[1][:7]):
        torch.testing.assert_close(a_, b_, rtol=2e-3, atol=2e-4 * float(a_.abs().max()))
    assert torch.equal(state[0][7], state[1][7]) and torch.equal(state[0][8], state[1][8])


def test_pipelined_lbfgs_matches_single_batch(device):
    """Seed shards on separate HIP streams (optim/pipelined.py) take exactly the iterates of the
    one-batch optimiser: same best cost / action per seed after a graph replay of 6 iterations."""
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg, PipelinedLBFGS
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    cfg = CollisionRolloutCfg()
    seeds = 12
    ocfg = LBFGSOptCfg(num_problems=seeds, inner_iters=6, num_iters=12)
    nls = len(ocfg.line_search_scale)
    start = torch.as_tensor(start_configuration(model), device=device)
    bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])

    def make(batch):
        ro = CollisionRollout(kin, scene, batch, cfg)
        ro.update_start_state(start)
        return ro.cost_and_gradient

    x0 = torch.as_tensor(seed_knots(model, seeds, cfg.n_knots, seed=4), device=device)
    one = LBFGSOpt(ocfg, make(seeds * nls), cfg.n_knots, kin.num_dof, bounds, device)
    ref = one.optimize(x0).clone()
    ref_cost = one.best_cost.clone()
    for shards in (2, 3, 4):
        pipe = PipelinedLBFGS(ocfg, make, cfg.n_knots, kin.num_dof, bounds, device, n_shards=shards)
        got = pipe.optimize(x0)
        torch.cuda.synchronize()
        assert torch.equal(pipe.best_cost, ref_cost), shards
        assert torch.equal(got, ref), shards
    assert float(ref_cost.min()) < 1e9


def test_rollout_protocol_members(device):
    """reference Rollout protocol (rollout_protocol.py): bounds, dt, metrics from an action,
    parameter updates that keep captured buffers valid"""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    B = 16
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg())
    assert ro.action_dim == 7 and ro.action_horizon == 12 and ro.sum_horizon and ro.dt == ro.cfg.traj_dt
    assert torch.equal(ro.action_bound_lows, kin.joint_limits_position[0])
    start = torch.as_tensor(start_configuration(model), device=device)
    assert ro.update_params(start_position=start)
    x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=device)
    m = ro.compute_metrics_from_action(x)
    assert float(m["scene_collision_cost"].max()) > 0.0
    cost = ro.cost_and_gradient(x.reshape(B, -1))[0].clone()
    # kernel sequence (metrics) vs fused launch: equal up to summation order, except on trajectories with
    # stationary points where the swept cost is discontinuous
    close = (m["cost"] - cost).abs() <= 2e-5 * cost.abs() + 1e-3
    assert float(close.float().mean()) >= 0.75, (m["cost"], cost)
    torch.testing.assert_close(m["self_collision_cost"] + m["scene_collision_cost"], m["cost"], rtol=1e-5, atol=1e-3)
    assert m["feasible"].dtype == torch.bool and bool((m["feasible"] == (m["cost"] == 0)).all()) and m["position"].shape == (B, 33, 7)
    # a moved world changes the costs without new buffers
    arrays = cuboid_scene_arrays([[{"dims": [0.2, 0.2, 0.2], "pose": [3.0, 3.0, 3.0, 1, 0, 0, 0]}] * 4])
    ptr = ro.cost.data_ptr()
    ro.update_params(scene=SceneData.from_arrays(arrays, device))
    c2, _ = ro.cost_and_gradient(x.reshape(B, -1))
    assert ro.cost.data_ptr() == ptr
    torch.testing.assert_close(c2, m["self_collision_cost"], rtol=2e-5, atol=1e-3)  # only self collision is left
    assert float(ro.compute_metrics_from_action(x)["scene_collision_cost"].abs().max()) == 0.0
