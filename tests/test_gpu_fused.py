"""Parity of the fused rollout kernel (csrc/rollout_fused.hip) against the drop-in kernel
sequence (same device functions, different summation order) and against the oracle."""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def _pair(device, robot="franka", seeds=24, world=None, voxel=False, **cfg_kw):
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model(robot)
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(world if world is not None else c2_world())
    if voxel:  # cuboids + one ESDF grid (a sphere obstacle) in the same scene
        from curobo_amd.scene import voxel_grid_from_sdf

        def sdf(p):
            return np.linalg.norm(p - np.array([0.35, -0.3, 0.5]), axis=-1) - 0.18

        arrays = {**arrays, **voxel_grid_from_sdf(sdf, (40, 40, 40), 0.04, pose7=(0.1, -0.1, 0.5, 1, 0, 0, 0),
                                                  max_distance=100.0)}
    scene = SceneData.from_arrays(arrays, device)
    knots = seed_knots(model, seeds, cfg_kw.get("n_knots", 12), seed=11)
    start = start_configuration(model)
    ros = []
    for fused in (False, True):
        cfg = CollisionRolloutCfg(use_fused=fused, fused_materialize=True, **cfg_kw)
        ro = CollisionRollout(kin, scene, seeds, cfg)
        ro.update_start_state(torch.as_tensor(start, device=device))
        ros.append(ro)
    return model, arrays, knots, start, ros[0], ros[1]


def _compare(ro_ref, ro_fused, knots, device, moving_only=False):
    b = knots.shape[0]
    x = torch.as_tensor(knots, device=device).reshape(b, -1)
    assert ro_fused.fused_available()
    c1, g1 = [t.clone() for t in ro_fused.cost_and_gradient(x)]
    act = x.view(b, ro_ref.cfg.n_knots, -1)
    ro_ref.compute_kinematics(ro_ref.compute_state_from_action(act))
    torch.cuda.synchronize()
    # the two paths are compiled separately, so FMA contraction may differ in the last bit
    torch.testing.assert_close(ro_fused.position, ro_ref.position, rtol=0, atol=1e-6)
    torch.testing.assert_close(ro_fused.robot_spheres, ro_ref.robot_spheres, rtol=0, atol=1e-6)
    # The swept cost is discontinuous in the sphere positions at zero motion (see
    # test_gpu_rollout.py), so the kernel sequence is evaluated on the fused kernel's own
    # materialised spheres: identical inputs -> identical branches -> only summation order differs.
    ro_ref.robot_spheres.copy_(ro_fused.robot_spheres)
    c0 = ro_ref.compute_costs().clone()
    g0 = ro_ref.backward().clone().view(b, -1)
    torch.cuda.synchronize()
    assert float(c0.max()) > 0.0
    return c0.cpu().numpy(), g0.cpu().numpy(), c1.cpu().numpy(), g1.cpu().numpy()


@pytest.mark.parametrize("sweep,speed", [(False, False), (True, False), (True, True)])
def test_fused_equals_kernel_sequence_franka(device, sweep, speed):
    _, _, knots, _, ro_ref, ro_fused = _pair(device, use_sweep=sweep, use_speed_metric=speed)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    # identical sphere positions -> identical branches; only fp32 summation order differs
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


@pytest.mark.parametrize("use_self,use_scene", [(True, False), (False, True)])
def test_fused_single_cost_terms(device, use_self, use_scene):
    _, _, knots, _, ro_ref, ro_fused = _pair(device, use_self_collision=use_self, use_scene_collision=use_scene)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


@pytest.mark.parametrize("degree,n_knots,interp", [(4, 10, 2), (5, 8, 3), (3, 4, 8)])
def test_fused_bspline_degrees_and_horizons(device, degree, n_knots, interp):
    """padded horizons 31, 43 and 65: more points than 16-lane groups exercises the group loop"""
    _, _, knots, _, ro_ref, ro_fused = _pair(device, seeds=9, n_knots=n_knots, interpolation_steps=interp,
                                             bspline_degree=degree)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


def test_fused_cuboids_and_voxel_grid(device):
    _, _, knots, _, ro_ref, ro_fused = _pair(device, voxel=True)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


def test_fused_ur10e(device):
    _, _, knots, _, ro_ref, ro_fused = _pair(device, robot="ur10e", seeds=7)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


def test_fused_matches_oracle_discrete(oracle, device):
    """strict end-to-end check against the all-oracle pipeline (continuous, non-swept cost)"""
    from oracle.rollout_ref import rollout_cost_and_gradient

    model, arrays, knots, start, _, ro = _pair(device, use_sweep=False, use_speed_metric=False)
    ref = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start, use_sweep=False,
                                    use_speed_metric=False)
    cost, grad = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1))
    torch.cuda.synchronize()
    assert (ref["cost"] > 0).mean() > 0.5
    np.testing.assert_allclose(cost.cpu().numpy(), ref["cost"], rtol=1e-4, atol=1e-2)
    gk = ref["grad_knots"].reshape(knots.shape[0], -1)
    np.testing.assert_allclose(grad.cpu().numpy(), gk, rtol=2e-3, atol=2e-5 * np.abs(gk).max())


def test_fused_is_deterministic_and_stateless(device):
    _, _, knots, _, _, ro = _pair(device)
    x = torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1)
    c1, g1 = [t.clone() for t in ro.cost_and_gradient(x)]
    ro.cost_and_gradient(x * 0.5)
    c2, g2 = ro.cost_and_gradient(x)
    assert torch.equal(c1, c2) and torch.equal(g1, g2)


def test_longest_first_dispatch_changes_only_the_order(device):
    """The dispatch workspace maps workgroups to trajectories longest-first from measured durations:
    outputs stay bit-identical to blockIdx order over a sequence of launches, both order arrays stay
    permutations, and after a few launches the order follows the measured ticks (descending buckets)."""
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    B = 700  # more workgroups than fit on the chip at once, not a multiple of anything
    x = torch.as_tensor(seed_knots(model, B, 12, seed=5), device=device).reshape(B, -1)
    ros = []
    for lf in (False, True):
        ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_sweep=True, use_speed_metric=True,
                                                                 longest_first_dispatch=lf))
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        ros.append(ro)
    for it in range(5):
        xi = x * (1.0 - 0.02 * it)
        c0, g0 = [t.clone() for t in ros[0].cost_and_gradient(xi)]
        c1, g1 = ros[1].cost_and_gradient(xi)
        assert torch.equal(c0, c1) and torch.equal(g0, g1), f"launch {it}"
        ws = ros[1]._dispatch.ws.view(4, B).cpu().numpy()
        for p in (0, 1):
            assert np.array_equal(np.sort(ws[p]), np.arange(B)), f"order[{p}] is not a permutation after launch {it}"
    assert ros[0]._dispatch is None
    ticks = ws[2:4]
    assert (ticks > 0).all()  # every workgroup measured itself in both phases
    # the order built last was sorted from the other phase's ticks: bucket index (64 buckets) non-increasing
    last_phase = ros[1]._dispatch._phase
    src = ticks[1 - last_phase].astype(np.int64)
    order = ws[1 - last_phase]
    buckets = src[order] * 64 // (src.max() + 1)
    assert (np.diff(buckets) <= 0).all()


def test_fused_rejects_what_does_not_fit(device):
    """G1 humanoid (674 spheres, 162k pairs) exceeds the per-trajectory LDS budget: the entry point
    must say so (ValueError, like every argument error) and the rollout must fall back."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg

    model = load_model("unitree_g1")
    kin = KinematicsParams.from_model(model, device)
    ro = CollisionRollout(kin, None, 2, CollisionRolloutCfg(use_fused=True))
    assert not ro.fused_available()
    with pytest.raises(ValueError, match="LDS"):
        ro.cost_and_gradient_fused(torch.zeros(2, 12, kin.num_dof, device=device))
    cost, grad = ro.cost_and_gradient(torch.zeros(2, 12 * kin.num_dof, device=device))  # falls back
    torch.cuda.synchronize()
    assert torch.isfinite(cost).all() and torch.isfinite(grad).all()


def test_c5_shape_multi_env_horizon_64(oracle, device):
    """BASELINE config 5 in miniature: 16 problems, each with its OWN world (num_envs = 16,
    env_query_idx per trajectory; cuboids everywhere, one environment also carries an ESDF grid...
    here cuboid-only per env with different layouts), horizon 64 (12 knots x 4 interpolation
    steps, padded 65 -> 1024-thread workgroups with a leftover point).  Fused kernel vs the kernel
    sequence vs the all-oracle pipeline (non-swept for the strict end-to-end check)."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import seed_knots, start_configuration
    from oracle.rollout_ref import rollout_cost_and_gradient

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    rng = np.random.default_rng(5)
    envs = []
    for e in range(16):
        obs = [{"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}]
        for _ in range(1 + e % 3):
            p = rng.uniform([-0.6, -0.6, 0.2], [0.6, 0.6, 0.9])
            obs.append({"dims": list(rng.uniform(0.1, 0.35, size=3)), "pose": [*p, 1, 0, 0, 0], "enable": True})
        envs.append(obs)
    arrays = cuboid_scene_arrays(envs)
    scene = SceneData.from_arrays(arrays, device)
    seeds_per_problem, n_prob = 3, 16
    B = n_prob * seeds_per_problem
    env_idx = np.repeat(np.arange(n_prob, dtype=np.int32), seeds_per_problem)
    knots = seed_knots(model, B, 12, seed=21)
    start = start_configuration(model)
    outs = []
    for fused in (False, True):
        cfg = CollisionRolloutCfg(interpolation_steps=4, use_sweep=False, use_speed_metric=False, use_fused=fused)
        ro = CollisionRollout(kin, scene, B, cfg)
        ro.update_start_state(torch.as_tensor(start, device=device))
        ro.update_env_query_idx(torch.as_tensor(env_idx, device=device))
        if fused:
            assert ro.fused_available() and cfg.padded_horizon == 65
        c, g = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(B, -1))
        torch.cuda.synchronize()
        outs.append((c.cpu().numpy().copy(), g.cpu().numpy().copy()))
    ref = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start, interpolation_steps=4,
                                    use_sweep=False, use_speed_metric=False, env_query_idx=env_idx)
    assert (ref["cost"] > 0).mean() > 0.5
    # different worlds must matter: the same seeds against env 0 only give a different answer
    ref0 = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start, interpolation_steps=4,
                                     use_sweep=False, use_speed_metric=False)
    assert np.abs(ref0["cost"] - ref["cost"]).max() > 1.0
    gk = ref["grad_knots"].reshape(B, -1)
    for c, g in outs:
        np.testing.assert_allclose(c, ref["cost"], rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(g, gk, rtol=2e-3, atol=2e-5 * np.abs(gk).max())


def test_fused_many_obstacles_uses_the_row_pass(device):
    """More than 32 obstacle records do not fit the packed scene pass's entry format (and the link
    masks only cover 32): the kernel falls back to one sphere per lane; same numbers as the sequence."""
    from curobo_amd.workloads import c2_world

    rng = np.random.default_rng(4)
    world = c2_world()[0]
    for _ in range(33):  # small boxes scattered through the workspace
        p = rng.uniform([-0.6, -0.6, 0.1], [0.7, 0.7, 0.9])
        world.append({"dims": rng.uniform(0.04, 0.12, 3).tolist(), "pose": [*p.tolist(), 1, 0, 0, 0]})
    assert len(world) == 37
    # discrete collision: with ~9x the (sphere, obstacle) sweep evaluations of the C2 world, the swept cost's
    # zero-motion discontinuity would make two separately compiled paths disagree on
    # the odd trajectory; the fallback pass itself is the same code for both modes
    _, _, knots, _, ro_ref, ro_fused = _pair(device, seeds=12, world=[world], use_sweep=False, use_speed_metric=False)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


def test_fused_every_sphere_in_collision(device):
    """Three large slabs through the workspace: every sphere of every point is in (swept) collision
    with several obstacles, so the packed scene pass runs with full rings (wrap-around, evaluation
    rounds in the middle of a sphere's obstacle list); same numbers as the kernel sequence."""
    big = [[{"dims": [3.0, 3.0, 0.4], "pose": [0, 0, 0.5, 1, 0, 0, 0]},
            {"dims": [0.5, 3.0, 3.0], "pose": [0.3, 0, 0.5, 1, 0, 0, 0]},
            {"dims": [3.0, 0.5, 3.0], "pose": [0, 0.2, 0.5, 0.9238795, 0, 0, 0.3826834]}]]
    for kw in (dict(), dict(use_sweep=False, use_speed_metric=False)):
        _, _, knots, _, ro_ref, ro_fused = _pair(device, seeds=12, world=big, **kw)
        c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
        np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
        np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())


def test_fused_swept_matches_oracle_at_c2_size(oracle, device):
    """The BENCHMARKED mode (fused launch, swept scene collision + speed metric + self collision) against the
    oracle directly, at the C2 size (256 seeds x 4 line-search candidates = 1024 trajectories x 33 points).

    (1) Same inputs, EVERY trajectory: the oracle's collision stages run on the spheres the fused launch materialises,
        its obstacle-frame transform in the device's arithmetic (sweep_allowance.device_frame_arithmetic: the sweep's
        duplicate centre sample exists iff half_dist > 0, wp_sweep_collision_kernel.py:186-203, a last-bit decision for a
        sphere that is stationary up to rounding -- the seeds come to rest at the last knot, about a third of them inside
        an obstacle's activation shell); its VJP on FK of the fused launch's own joint positions.  Cost 1e-5, gradient 5e-4.
    (2) Same inputs, the oracle in the REFERENCE's arithmetic, per sphere (the kernel sequence's scene kernel on the fused
        launch's spheres): every sphere without a stationary neighbour tight, a stationary one differs by whole
        centre-sample terms only; the fused launch per trajectory = oracle + exactly those terms, 1e-5.
    (3) All-oracle pipeline from the knots (its own FK, 1e-6 m away from the device's): trajectories without a resting
        colliding sphere tightly, the others inside the 3x band."""
    from sweep_allowance import device_frame_arithmetic, per_sphere_allowance, rest_in_collision

    from curobo_amd.workloads import seed_knots
    from oracle.rollout_ref import rollout_cost_and_gradient

    seeds, nls = 256, 4
    model, arrays, _, start, seq, ro = _pair(device, seeds=seeds * nls)
    cfg = ro.cfg
    assert cfg.use_sweep and cfg.use_speed_metric and cfg.padded_horizon == 33
    base = seed_knots(model, seeds, cfg.n_knots, seed=2)
    # the four candidates of a seed, as the line search spreads them along a descent direction
    rng = np.random.default_rng(0)
    step = rng.normal(size=base.shape).astype(np.float32) * 0.02
    knots = np.stack([base + a * step for a in (0.0, 0.1, 0.5, 1.0)], axis=1).reshape(seeds * nls, cfg.n_knots, -1)
    B, nk, D = knots.shape
    x = torch.as_tensor(knots, device=device).reshape(B, -1)
    cost, grad = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    cost, grad = cost.cpu().numpy(), grad.cpu().numpy().reshape(B, nk, D)
    md, ph, S = model.as_dict(), cfg.padded_horizon, model.num_spheres
    w, eta = cfg.scene_collision_weight, cfg.activation_distance
    # ---- (1) oracle stages on the fused launch's own materialised state, device arithmetic of the frame transform
    sph = ro.robot_spheres.cpu().numpy()
    pos = ro.position.cpu().numpy()
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, cfg.self_collision_weight)
    with device_frame_arithmetic(oracle):
        wc = oracle.scene_collision(sph, arrays, w, eta, sweep=True, enable_speed_metric=True, speed_dt=cfg.traj_dt)
    ref_cost = oracle.trajectory_cost_sum(sc["distance"].reshape(B, ph), wc["distance"])
    assert (ref_cost > 0).mean() > 0.5 and (wc["distance"] > 0).sum() > 10000 and (sc["distance"] > 0).sum() > 100
    amb1 = rest_in_collision(sph, wc["distance"])
    err = np.abs(cost - ref_cost) / (1e-5 * np.abs(ref_cost) + 1e-7 * w)
    msg = (f"[c2] {int(amb1.sum())} of {B} trajectories ({amb1.mean():.3f}) hold a sphere that rests in collision; worst cost error "
           f"{float(err.max()):.3f} of the 1e-5 bound (resting {float(err[amb1].max()):.3f}, moving {float(err[~amb1].max()):.3f}); "
           f"beyond the bound: {int((err > 1).sum())}")
    print("\n" + msg)
    assert amb1.sum() > 50, "the case the device arithmetic exists for must be in the sample"
    assert (err <= 1.0).all(), msg
    fk = oracle.kinematics_forward(pos.reshape(B * ph, D), md, horizon=ph)
    np.testing.assert_allclose(sph.reshape(B * ph, S, 4), fk["robot_spheres"], atol=1e-5)  # north_star: FK within 1e-5
    gs = sc["gradient"].reshape(B, ph, S, 4).copy()
    gs[..., :3] += wc["gradient"][..., :3]
    gq = oracle.kinematics_backward(md, fk["cumul_mat"], gs.reshape(B * ph, S, 4), horizon=ph)
    z = np.zeros((B, ph, D), np.float32)
    gk = oracle.bspline_backward(gq.reshape(B, ph, D), z, z, z, np.array([cfg.traj_dt], np.float32), np.zeros(B, np.int32),
                                 np.zeros(1, np.uint8), nk, cfg.bspline_degree)
    np.testing.assert_allclose(grad, gk, rtol=5e-4, atol=5e-6 * np.abs(gk).max(), err_msg=msg)
    # ---- (2) the reference's arithmetic, per sphere: the kernel sequence's scene kernel on the SAME spheres
    seq.compute_kinematics(seq.compute_state_from_action(x.view(B, nk, D)))
    seq.robot_spheres.copy_(ro.robot_spheres)
    seq.compute_costs()
    torch.cuda.synchronize()
    d, g = seq.scene_dist.cpu().numpy(), seq.scene_grad.cpu().numpy()[..., :3]
    wr = oracle.scene_collision(sph, arrays, w, eta, sweep=True, enable_speed_metric=True, speed_dt=cfg.traj_dt)
    in_col = wr["distance"] > 0
    # (one cuboid of the C2 world is rotated: R as a matrix here, the quaternion form in the oracle; see the C5 test)
    res = per_sphere_allowance(oracle, d, g, wr["distance"], wr["gradient"][..., :3], sph, arrays, w, eta, None, cfg.traj_dt,
                               max(3, int(2e-5 * in_col.sum())), "c2", tol_abs=2e-6, min_colliding=10000)
    want = (wr["distance"].astype(np.float64).sum((1, 2)) + res["corr"].sum((1, 2))
            + sc["distance"].reshape(B, -1).astype(np.float64).sum(1))
    n_col = np.maximum(in_col.sum((1, 2)), 1)
    e_c = np.abs(cost - want) / (np.abs(want) + 1e-2 * w * n_col)
    keep = ~res["split"].any((1, 2))
    print(f"[c2] fused launch vs oracle (reference arithmetic) + whole-term corrections, per trajectory: max relative cost error "
          f"{float(e_c.max()):.2e}; ambiguous sphere fraction {res['frac_amb']:.2e}, flipped {res['flipped']}")
    assert (e_c[keep] <= 1e-5).all(), f"{int((e_c[keep] > 1e-5).sum())} trajectories beyond 1e-5 (ambiguous sphere fraction {res['frac_amb']:.2e})"
    # ---- (3) the all-oracle pipeline from the knots

    def in_band(a, b):
        return (a <= 3.001 * b + 1e-3 * w) & (b <= 3.001 * a + 1e-3 * w)

    ref = rollout_cost_and_gradient(oracle, md, arrays, knots, start)
    ambiguous = rest_in_collision(ref["robot_spheres"], ref["scene_cost"]) | amb1
    clean = ~ambiguous
    assert clean.sum() >= 0.5 * B, f"only {clean.sum()} of {B} trajectories are free of stationary colliding spheres"
    rel = np.abs(cost - ref["cost"]) / np.maximum(np.abs(ref["cost"]), 1e-3 * w)
    # FK rounding differs between the two pipelines (1e-6 m on a sphere = 1e-6 * w on its cost): 1e-5 of the
    # weight scale per trajectory; a handful of trajectories may still flip a sweep `break` comparison
    print(f"[c2] all-oracle pipeline from the knots: {int(ambiguous.sum())} of {B} trajectories ({ambiguous.mean():.3f}) held to the 3x band "
          f"only (own FK -> own last bits); clean: median {float(np.median(rel[clean])):.2e}, 99 % {float(np.quantile(rel[clean], 0.99)):.2e}")
    assert np.quantile(rel[clean], 0.99) < 1e-4 and np.median(rel[clean]) < 1e-5, np.sort(rel[clean])[-5:]
    assert in_band(cost[ambiguous], ref["cost"][ambiguous]).all()
    gkr = ref["grad_knots"].reshape(B, nk, D)
    gerr = np.abs(grad - gkr).reshape(B, -1).max(-1) / np.abs(gkr).max()
    assert np.quantile(gerr[clean], 0.99) < 2e-3, np.sort(gerr[clean])[-5:]


def test_fused_world_with_analytic_primitives(oracle, device):
    """BASELINE config 2 says "sphere + cuboid world": sphere / capsule / cylinder obstacles as analytic records of
    the cuboid store, fused launch vs the kernel sequence and vs the all-oracle pipeline (non-swept: continuous)."""
    from test_scene_primitives import PRIM_WORLD

    from oracle.rollout_ref import rollout_cost_and_gradient

    _, _, knots, _, ro_ref, ro_fused = _pair(device, world=PRIM_WORLD)
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())
    model, arrays, knots, start, _, ro = _pair(device, world=PRIM_WORLD, use_sweep=False, use_speed_metric=False)
    ref = rollout_cost_and_gradient(oracle, model.as_dict(), arrays, knots, start, use_sweep=False, use_speed_metric=False)
    cost, grad = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1))
    torch.cuda.synchronize()
    assert (ref["scene_cost"] > 0).mean() > 0.01
    np.testing.assert_allclose(cost.cpu().numpy(), ref["cost"], rtol=1e-4, atol=1e-2)
    gk = ref["grad_knots"].reshape(knots.shape[0], -1)
    np.testing.assert_allclose(grad.cpu().numpy(), gk, rtol=2e-3, atol=2e-5 * np.abs(gk).max())


@pytest.mark.parametrize("robot,n_left", [("franka", 1), ("ur10e", 1), ("franka", 0)])
def test_self_collision_lane_lists_change_no_bit(device, robot, n_left):
    """the lane = sphere form of the self-collision pass (pair list dealt to lanes, curobo_hip_self_lane_lists_host) against
    the row form that walks pair_locations: same maxima, same arg-max pair -> bit-identical cost and gradient; with and
    without a leftover point (padded horizon 33 on 32 rows; 31 on 32 rows)"""
    kw = dict(use_scene_collision=False) if n_left else dict(use_scene_collision=False, n_knots=10, interpolation_steps=2, bspline_degree=4)
    model, _, knots, _, _, ro = _pair(device, robot=robot, seeds=24, **kw)
    pairs = ro.kin.self_collision.collision_pairs
    lanes = getattr(pairs, "_self_lane_lists", None)
    assert lanes is not None, "KinematicsParams.from_model deals the pair list to lanes"
    code = lanes[1]
    assert ((code & 0xffff) + (code >> 16)) * 64 == lanes[0].numel()
    x = torch.as_tensor(knots, device=device).reshape(knots.shape[0], -1)
    # push the arm into itself so that many points are in self collision
    x = x * 1.6
    c1, g1 = [t.clone() for t in ro.cost_and_gradient(x)]
    del pairs._self_lane_lists
    try:
        c0, g0 = [t.clone() for t in ro.cost_and_gradient(x)]
    finally:
        pairs._self_lane_lists = lanes
    torch.cuda.synchronize()
    assert float((c0 > 0).float().mean()) > 0.3
    assert torch.equal(c0, c1) and torch.equal(g0, g1)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fused_equals_the_sequence_on_randomly_rotated_cuboids(device, seed):
    """Random worlds of 3 .. 10 cuboids with random rotations (the fixed worlds of this file rotate about one axis: their
    rotation matrices are built from exact products).  The two paths must hold the SAME obstacle-frame rotation, bit for bit:
    a sphere that rests in collision up to rounding otherwise gets the sweep's duplicate centre sample in one path only
    (scene_device.hpp::load_rec_global; found by tests/randomised/fuzz_fused.py).  Every trajectory is compared, the resting ones too."""
    rng = np.random.default_rng(seed)
    world = []
    for _ in range(int(rng.integers(3, 11))):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        world.append({"dims": [float(v) for v in rng.uniform(0.05, 0.6, size=3)],
                      "pose": [float(v) for v in rng.uniform([-0.7, -0.7, -0.2], [0.7, 0.7, 1.0])] + [float(v) for v in q]})
    _, _, knots, _, ro_ref, ro_fused = _pair(device, seeds=48, world=[world])
    c0, g0, c1, g1 = _compare(ro_ref, ro_fused, knots, device)
    p = ro_fused.robot_spheres.cpu().numpy()[..., :3]
    still = np.linalg.norm(np.diff(p, axis=1), axis=-1) < 1e-5
    resting = (still & (ro_ref.scene_dist.cpu().numpy().reshape(p.shape[:3])[:, 1:] > 0)).any(axis=(1, 2))
    assert resting.sum() >= 5, "the inputs must contain trajectories with a sphere that rests in collision"
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-3, atol=2e-5 * np.abs(g0).max())
