"""Oracle-direct parity of the configurations ``bench.py`` measures beyond C2:

* C5: the multi-environment launch over worlds that hold cuboids AND an fp16 ESDF grid each, horizon 64 (padded 65),
  swept scene collision + speed metric + self collision -- ``rollout_trajectory_fused_kernel<3, 3, 3>``;
* C3: UR10e through the 128^3 ESDF at the benchmark's size (512 seeds x 4 candidates x 33 points): the SWEEP x voxel
  instantiation of the fused launch and ``scene_collision_packed_kernel`` of the kernel sequence.

Protocol of ``test_gpu_fused.py::test_fused_swept_matches_oracle_at_c2_size`` part (1): the oracle's collision stages run
on the spheres the launch itself materialises, WITH THE DEVICE'S ARITHMETIC for the world -> obstacle-frame transform
(``sweep_allowance.device_frame_arithmetic``), so that the sweep's `half_dist > 0` decision of a sphere that is stationary up
to rounding is the device's: EVERY trajectory is held to cost 1e-5 / gradient 5e-4.  The per-sphere tests hold the kernels
to the oracle in the REFERENCE's arithmetic with the whole-centre-term allowance per stationary sphere.
"""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


from sweep_allowance import device_frame_arithmetic, per_sphere_allowance, rest_in_collision as _rest_in_collision  # noqa: E402


def _candidates(model, seeds, nls, n_knots, seed):
    """``nls`` line-search candidates per seed, spread along a random direction as the optimiser spreads them"""
    from curobo_amd.workloads import seed_knots

    base = seed_knots(model, seeds, n_knots, seed=seed)
    rng = np.random.default_rng(seed)
    step = rng.normal(size=base.shape).astype(np.float32) * 0.02
    alphas = (0.0, 0.1, 0.5, 1.0)[:nls]
    return np.stack([base + a * step for a in alphas], axis=1).reshape(seeds * nls, n_knots, -1)


def _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, env_idx, cost, grad, tag="fused"):
    """cost [B], grad [B, nk, D] of a fused launch (``fused_materialize``) vs the oracle's stages on the launch's own
    spheres / joint positions, the oracle's obstacle-frame transform in the device's arithmetic: EVERY trajectory at
    cost 1e-5 / gradient 5e-4, the ones with spheres that rest in collision too.  Returns (resting mask, self, scene)."""
    cfg = ro.cfg
    B, nk, D = knots.shape
    md, ph, S = model.as_dict(), cfg.padded_horizon, model.num_spheres
    sph = ro.robot_spheres.cpu().numpy()
    pos = ro.position.cpu().numpy()
    multi = env_idx is not None
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, cfg.self_collision_weight)
    with device_frame_arithmetic(oracle):
        wc = oracle.scene_collision(sph, arrays, cfg.scene_collision_weight, cfg.activation_distance, sweep=True,
                                    enable_speed_metric=True, speed_dt=cfg.traj_dt, env_query_idx=env_idx, use_multi_env=multi)
    ref_cost = oracle.trajectory_cost_sum(sc["distance"].reshape(B, ph), wc["distance"])
    w = cfg.scene_collision_weight
    amb = _rest_in_collision(sph, wc["distance"])
    err = np.abs(cost - ref_cost) / (1e-5 * np.abs(ref_cost) + 1e-7 * w)
    msg = (f"[{tag}] {int(amb.sum())} of {B} trajectories ({amb.mean():.3f}) hold a sphere that rests in collision; worst cost error "
           f"{float(err.max()):.3f} of the bound (resting: {float(err[amb].max()) if amb.any() else 0.0:.3f}, "
           f"moving: {float(err[~amb].max()) if (~amb).any() else 0.0:.3f}); beyond the bound: {int((err > 1).sum())}")
    print("\n" + msg)
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
    return amb, sc, wc


def test_c5_bench_path_multi_env_cuboids_and_esdf_matches_oracle(oracle, device):
    """What ``bench.py`` C5 launches: two planning problems, each in its own world of cuboids + one 64^3 fp16 ESDF grid,
    64 seeds x 4 candidates per problem, 12 knots x 4 interpolation steps (padded horizon 65), swept + speed metric."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c5_mixed_worlds, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    n_prob, seeds, nls = 2, 64, 4
    B = n_prob * seeds * nls
    arrays = c5_mixed_worlds(n_prob, voxels=True)
    assert arrays["voxel_features"].shape[0] == n_prob and arrays["cuboid_dims"].shape[0] == n_prob
    scene = SceneData.from_arrays(arrays, device)
    assert scene.struct.max_cuboids >= 2 and scene.struct.max_voxel_grids == 1 and scene.struct.voxel_coarse_min is not None
    cfg = CollisionRolloutCfg(interpolation_steps=4, use_fused=True, fused_materialize=True)
    assert cfg.padded_horizon == 65 and cfg.use_sweep and cfg.use_speed_metric
    ro = CollisionRollout(kin, scene, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    env_idx = np.repeat(np.arange(n_prob, dtype=np.int32), seeds * nls)
    ro.update_env_query_idx(torch.as_tensor(env_idx, device=device))
    assert ro.fused_available() and ro.use_multi_env
    knots = _candidates(model, n_prob * seeds, nls, cfg.n_knots, seed=8)
    cost, grad = ro.cost_and_gradient(torch.as_tensor(knots, device=device).reshape(B, -1))
    torch.cuda.synchronize()
    cost, grad = cost.cpu().numpy(), grad.cpu().numpy().reshape(knots.shape)
    resting, sc, wc = _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, env_idx, cost, grad, tag="c5 small")
    assert resting.sum() >= 0.1 * B, "the case the device arithmetic exists for must be in the sample"
    # both obstacle kinds and both worlds take part: hits against the ESDF-only and the cuboid-only version of the worlds
    sph = ro.robot_spheres.cpu().numpy()
    only_vox = {k: v for k, v in arrays.items() if k.startswith("voxel")}
    only_cub = {k: v for k, v in arrays.items() if k.startswith("cuboid")}
    for part in (only_vox, only_cub):
        d = oracle.scene_collision(sph, part, 1.0, cfg.activation_distance, env_query_idx=env_idx, use_multi_env=True)["distance"]
        for e in range(n_prob):
            assert (d[env_idx == e] > 0).sum() > 50
    other = oracle.scene_collision(sph, arrays, 1.0, cfg.activation_distance, env_query_idx=1 - env_idx, use_multi_env=True)["distance"]
    mine = oracle.scene_collision(sph, arrays, 1.0, cfg.activation_distance, env_query_idx=env_idx, use_multi_env=True)["distance"]
    assert np.abs(other - mine).max() > 1e-2, "the two worlds must differ where the robot is"


def test_c3_size_sweep_x_voxel_fused_and_packed_kernel_match_oracle(oracle, device):
    """BASELINE config 3 at the benchmark's size: UR10e, one 128^3 fp16 ESDF grid at 0.02 m, 512 seeds x 4 candidates x
    33 points = 67 584 points.  (1) the fused launch (SWEEP x voxel instantiation), (2) ``scene_collision_packed_kernel``
    -- the swept voxel scene kernel of the kernel sequence, with the coarse culling grid -- per sphere."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c3_voxel_world, start_configuration

    model = load_model("ur10e")
    kin = KinematicsParams.from_model(model, device)
    arrays = c3_voxel_world()
    assert arrays["voxel_features"].shape[-1] == 128 ** 3
    scene = SceneData.from_arrays(arrays, device)
    seeds, nls = 512, 4
    B = seeds * nls
    start = torch.as_tensor(start_configuration(model), device=device)
    knots = _candidates(model, seeds, nls, 12, seed=4)
    x = torch.as_tensor(knots, device=device).reshape(B, -1)
    # (1) fused
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_fused=True, fused_materialize=True))
    ro.update_start_state(start)
    assert ro.fused_available() and ro.cfg.padded_horizon == 33
    cost, grad = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    cost, grad = cost.cpu().numpy(), grad.cpu().numpy().reshape(knots.shape)
    # (the benchmark's start configuration touches the ESDF's activation shell and a B-spline that starts from rest
    # repeats its first points: every trajectory holds stationary spheres in collision; the grid frame is a translation)
    assert np.allclose(arrays["voxel_inv_pose"][0, 0, 3:7], [1, 0, 0, 0])
    resting, sc, wc = _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, None, cost, grad, tag="c3")
    assert resting.all() and (wc["distance"] > 0).sum() > 10000
    # (2) the kernel sequence's scene kernel on the SAME spheres, per sphere
    seq = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_fused=False))
    seq.update_start_state(start)
    seq.compute_kinematics(seq.compute_state_from_action(x.view(B, 12, -1)))
    seq.robot_spheres.copy_(ro.robot_spheres)
    seq.compute_costs()
    torch.cuda.synchronize()
    d, g = seq.scene_dist.cpu().numpy(), seq.scene_grad.cpu().numpy()
    sph = ro.robot_spheres.cpu().numpy()
    w = ro.cfg.scene_collision_weight
    assert np.array_equal(d > 0, wc["distance"] > 0), "which spheres collide must be identical"
    p = sph[..., :3]
    stepn = np.linalg.norm(np.diff(p, axis=1), axis=-1)
    moving = np.ones(p.shape[:3], bool)
    moving[:, 1:] &= stepn >= 1e-5
    moving[:, :-1] &= stepn >= 1e-5
    # per sphere the tolerance is the one of the kernel tests (test_gpu_kernels.py::test_scene_collision_voxels: 2e-5 of
    # the weight, i.e. 2e-5 m of penetration): up to seven trilinear samples per sphere are summed and the speed metric
    # scales the sum; measured on this workload: 8 ulp-level differences of 1e-6 m at most (tools/diag_r03.py)
    # held to 1e-5 relative + 1e-5 m of penetration per sphere (the kernel tests' bound is 1e-4 + 2e-5 m); measured on this
    # workload: the worst sphere sits at 1.21 x (1e-5 relative + 5e-6 m) -- one fp16-grid trilinear term of seven (round 6)
    err = np.abs(d[moving] - wc["distance"][moving])
    worst = float((err / (1e-5 * np.abs(wc["distance"][moving]) + 5e-6 * w)).max())
    print(f"[C3 per sphere] worst error in units of (1e-5 relative + 5e-6 m of penetration): {worst:.3f}")
    assert worst < 2.0, worst
    gerr = np.abs(g[moving][:, :3] - wc["gradient"][moving][:, :3])
    gtol = 1e-3 * np.abs(wc["gradient"][moving][:, :3]) + 2e-4 * w
    # (the speed metric divides by the sphere's speed: a handful of slow spheres carry a few 1e-3 of relative error)
    assert (gerr > gtol).mean() < 1e-5 and (gerr <= 30 * gtol).all(), (int((gerr > gtol).sum()), float((gerr / gtol).max()))
    rest = ~moving & (wc["distance"] > 0)
    assert rest.sum() > 1000
    np.testing.assert_allclose(d[rest], wc["distance"][rest], rtol=1e-4, atol=2e-5 * w)  # axis-aligned grid: no 1x / 2x / 3x split
    # per trajectory the scene cost agrees to the 1e-5 of the fused comparison
    np.testing.assert_allclose(d.sum((1, 2)), wc["distance"].sum((1, 2)), rtol=1e-5, atol=1e-7 * w)


@pytest.mark.parametrize("rotated", [False, True], ids=["bench_worlds", "rotated_cuboids"])
def test_c5_bench_size_per_sphere_sweep_allowance(rotated, oracle, device):
    """C5 at the SIZE ``bench.py`` runs per GPU share (2 problems x 512 seeds x 4 candidates = 4096 trajectories x 65
    points x 65 spheres), held to the oracle PER SPHERE.

    The reference's swept kernel duplicates the centre sample of a direction iff the half sweep length in the obstacle
    frame is > 0 (wp_sweep_collision_kernel.py:186-203): for a sphere that is stationary up to rounding, two
    implementations may land on different sides.  The allowance is therefore per (sphere, obstacle, direction): a sphere
    with n stationary neighbours may differ from the oracle by k_o centre-sample terms of obstacle o, |k_o| <= n, and by
    nothing else; every other sphere of the same trajectory is compared tightly.  (1) the kernel sequence's swept scene
    kernel on the fused launch's own spheres, per sphere; (2) the fused launch's per-trajectory cost against the sum of
    the oracle's per-sphere costs corrected by exactly those k_o terms, 1e-5.  ``bench_worlds`` are what ``bench.py``
    launches (axis-aligned obstacle frames: the transform into them is exact and no sphere needs the allowance --
    measured: 58 130 stationary colliding spheres, 0 whole-term corrections); ``rotated_cuboids`` turns the random cuboids
    so that the allowance is exercised at the same size."""
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c5_mixed_worlds, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    n_prob, seeds, nls = 2, 512, 4
    B = n_prob * seeds * nls
    arrays = c5_mixed_worlds(n_prob, voxels=True, rotated=rotated)
    scene = SceneData.from_arrays(arrays, device)
    cfg = CollisionRolloutCfg(interpolation_steps=4, use_fused=True, fused_materialize=True)
    ro = CollisionRollout(kin, scene, B, cfg)
    start = torch.as_tensor(start_configuration(model), device=device)
    ro.update_start_state(start)
    env_idx = np.repeat(np.arange(n_prob, dtype=np.int32), seeds * nls)
    ro.update_env_query_idx(torch.as_tensor(env_idx, device=device))
    assert ro.fused_available() and ro.use_multi_env and cfg.padded_horizon == 65
    knots = _candidates(model, n_prob * seeds, nls, cfg.n_knots, seed=8)
    x = torch.as_tensor(knots, device=device).reshape(B, -1)
    cost, _ = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    cost = cost.cpu().numpy().astype(np.float64)
    sph = ro.robot_spheres.cpu().numpy()
    # (1) the kernel sequence's scene kernel on the SAME spheres
    seq = CollisionRollout(kin, scene, B, CollisionRolloutCfg(interpolation_steps=4, use_fused=False))
    seq.update_start_state(start)
    seq.update_env_query_idx(torch.as_tensor(env_idx, device=device))
    seq.compute_kinematics(seq.compute_state_from_action(x.view(B, cfg.n_knots, -1)))
    seq.robot_spheres.copy_(ro.robot_spheres)
    seq.compute_costs()
    torch.cuda.synchronize()
    d, g = seq.scene_dist.cpu().numpy(), seq.scene_grad.cpu().numpy()[..., :3]
    w, eta = cfg.scene_collision_weight, cfg.activation_distance
    wc = oracle.scene_collision(sph, arrays, w, eta, sweep=True, enable_speed_metric=True, speed_dt=cfg.traj_dt,
                                env_query_idx=env_idx, use_multi_env=True)
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, cfg.self_collision_weight)
    d_ref, g_ref = wc["distance"], wc["gradient"][..., :3]
    in_col = d_ref > 0
    # The sweep has a second discontinuity of the same kind: it stops when the accumulated jump reaches the half segment
    # (`if jump >= half_dist: break`, wp_sweep_collision_kernel.py:197-203), so a sphere whose jump lands within rounding of
    # its half segment takes one sample more or less.  In an axis-aligned obstacle frame both implementations compute the
    # same bits; in a rotated one (R as a matrix here, the quaternion form in the oracle) about one colliding sphere in 1e5
    # is split.  Such spheres are counted, bounded, and their trajectories left out of the per-trajectory comparison.
    max_split = 0 if not rotated else max(3, int(2e-5 * in_col.sum()))
    res = per_sphere_allowance(oracle, d, g, d_ref, g_ref, sph, arrays, w, eta, env_idx, cfg.traj_dt, max_split, "c5", min_colliding=100000)
    corr, split, frac_amb = res["corr"], res["split"], res["frac_amb"]
    # (2) the fused launch, per trajectory: oracle per-sphere costs + exactly the allowed corrections + self collision
    want = d_ref.astype(np.float64).sum((1, 2)) + corr.sum((1, 2)) + sc["distance"].reshape(B, -1).astype(np.float64).sum(1)
    # (the criterion of _check_against_oracle_on_same_inputs: 1e-5 relative + 1e-7 m of penetration; in the rotated worlds
    # 1e-7 m PER COLLIDING SPHERE of the trajectory: R as a matrix here, the quaternion form in the oracle -- the obstacle-
    # frame coordinates of a sphere, and with them its signed distance, differ by an ulp of a ~1 m coordinate, 6e-8 m)
    n_col = np.maximum((d_ref > 0).sum((1, 2)), 1) if rotated else 1
    e_c = np.abs(cost - want) / (np.abs(want) + 1e-2 * w * n_col)
    seq_sum = d.astype(np.float64).sum((1, 2)) + seq.self_dist.cpu().numpy().reshape(B, -1).astype(np.float64).sum(1)
    e_hip = np.abs(cost - seq_sum) / (np.abs(seq_sum) + 1e-2 * w)
    iw = int(np.argmax(e_c))
    print(f"[c5 per sphere] fused launch vs oracle per trajectory: max relative cost error {float(e_c.max()):.2e} "
          f"({int((e_c > 1e-5).sum())} of {B} beyond 1e-5); fused vs the kernel sequence's own per-sphere sum {float(e_hip.max()):.2e}; "
          f"worst trajectory {iw}: fused {cost[iw]:.4f}, oracle + corrections {want[iw]:.4f}, kernel sequence {seq_sum[iw]:.4f}, "
          f"colliding spheres {int((d_ref[iw] > 0).sum())}, |corrections| {float(np.abs(corr[iw]).sum()):.4f}")
    keep = ~split.any((1, 2))
    assert (e_c[keep] <= 1e-5).all(), f"{int((e_c[keep] > 1e-5).sum())} trajectories beyond 1e-5 (ambiguous sphere fraction {frac_amb:.2e})"
