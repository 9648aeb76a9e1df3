"""Oracle-direct parity of the configurations ``bench.py`` measures beyond C2:

* C5: the multi-environment launch over worlds that hold cuboids AND an fp16 ESDF grid each, horizon 64 (padded 65),
  swept scene collision + speed metric + self collision -- ``rollout_trajectory_fused_kernel<3, 3, 3>``;
* C3: UR10e through the 128^3 ESDF at the benchmark's size (512 seeds x 4 candidates x 33 points): the SWEEP x voxel
  instantiation of the fused launch and ``scene_collision_packed_kernel`` of the kernel sequence.

Protocol of ``test_gpu_fused.py::test_fused_swept_matches_oracle_at_c2_size`` part (1): the oracle's collision stages run
on the spheres the launch itself materialises (identical inputs -> identical sweep branches), its VJP on FK of the
launch's own joint positions; cost 1e-5, gradient 5e-4.  Trajectories that hold a sphere which is stationary up to
rounding AND in collision are only held to the 3x band (the reference algorithm's own discontinuity).
"""

import numpy as np
import pytest
import torch

from conftest import load_model

pytestmark = pytest.mark.gpu


def _rest_in_collision(spheres, scene_cost):
    p = spheres[..., :3]
    stepn = np.linalg.norm(np.diff(p, axis=1), axis=-1)
    still = np.zeros(p.shape[:3], bool)
    still[:, 1:] |= stepn < 1e-5
    still[:, :-1] |= stepn < 1e-5
    return (still & (scene_cost > 0)).any(axis=(1, 2))


def _candidates(model, seeds, nls, n_knots, seed):
    """``nls`` line-search candidates per seed, spread along a random direction as the optimiser spreads them"""
    from curobo_amd.workloads import seed_knots

    base = seed_knots(model, seeds, n_knots, seed=seed)
    rng = np.random.default_rng(seed)
    step = rng.normal(size=base.shape).astype(np.float32) * 0.02
    alphas = (0.0, 0.1, 0.5, 1.0)[:nls]
    return np.stack([base + a * step for a in alphas], axis=1).reshape(seeds * nls, n_knots, -1)


def _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, env_idx, cost, grad, axis_aligned_world=False):
    """cost [B], grad [B, nk, D] of a fused launch (``fused_materialize``) vs the oracle's stages on the launch's own
    spheres / joint positions.  Returns the mask of trajectories compared tightly.  ``axis_aligned_world``: every
    obstacle frame is a pure translation of the world frame, so the transform into it is exact, two sphere positions
    coincide there iff they coincide in the world, and the stationary-sphere discontinuity cannot split the two
    implementations: every trajectory is compared tightly."""
    cfg = ro.cfg
    B, nk, D = knots.shape
    md, ph, S = model.as_dict(), cfg.padded_horizon, model.num_spheres
    sph = ro.robot_spheres.cpu().numpy()
    pos = ro.position.cpu().numpy()
    multi = env_idx is not None
    sc = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, cfg.self_collision_weight)
    wc = oracle.scene_collision(sph, arrays, cfg.scene_collision_weight, cfg.activation_distance, sweep=True,
                                enable_speed_metric=True, speed_dt=cfg.traj_dt, env_query_idx=env_idx, use_multi_env=multi)
    ref_cost = oracle.trajectory_cost_sum(sc["distance"].reshape(B, ph), wc["distance"])
    w = cfg.scene_collision_weight
    amb = _rest_in_collision(sph, wc["distance"])
    if axis_aligned_world:
        amb[:] = False
    assert amb.mean() < 0.6, f"{amb.sum()} of {B} trajectories rest inside an obstacle"
    np.testing.assert_allclose(cost[~amb], ref_cost[~amb], rtol=1e-5, atol=1e-7 * w)
    band = (cost[amb] <= 3.001 * ref_cost[amb] + 1e-3 * w) & (ref_cost[amb] <= 3.001 * cost[amb] + 1e-3 * w)
    assert band.all()
    fk = oracle.kinematics_forward(pos.reshape(B * ph, D), md, horizon=ph)
    np.testing.assert_allclose(sph.reshape(B * ph, S, 4), fk["robot_spheres"], atol=1e-5)  # north_star: FK within 1e-5
    gs = sc["gradient"].reshape(B, ph, S, 4).copy()
    gs[..., :3] += wc["gradient"][..., :3]
    gq = oracle.kinematics_backward(md, fk["cumul_mat"], gs.reshape(B * ph, S, 4), horizon=ph)
    z = np.zeros((B, ph, D), np.float32)
    gk = oracle.bspline_backward(gq.reshape(B, ph, D), z, z, z, np.array([cfg.traj_dt], np.float32), np.zeros(B, np.int32),
                                 np.zeros(1, np.uint8), nk, cfg.bspline_degree)
    np.testing.assert_allclose(grad[~amb], gk[~amb], rtol=5e-4, atol=5e-6 * np.abs(gk).max())
    return ~amb, sc, wc


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
    tight, sc, wc = _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, env_idx, cost, grad)
    assert tight.sum() >= 0.4 * B
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
    tight, sc, wc = _check_against_oracle_on_same_inputs(oracle, model, arrays, ro, knots, None, cost, grad, axis_aligned_world=True)
    assert tight.all() and (wc["distance"] > 0).sum() > 10000
    assert _rest_in_collision(ro.robot_spheres.cpu().numpy(), wc["distance"]).all()
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
    np.testing.assert_allclose(d[moving], wc["distance"][moving], rtol=1e-4, atol=2e-5 * w)
    gerr = np.abs(g[moving][:, :3] - wc["gradient"][moving][:, :3])
    gtol = 1e-3 * np.abs(wc["gradient"][moving][:, :3]) + 2e-4 * w
    # (the speed metric divides by the sphere's speed: a handful of slow spheres carry a few 1e-3 of relative error)
    assert (gerr > gtol).mean() < 1e-5 and (gerr <= 30 * gtol).all(), (int((gerr > gtol).sum()), float((gerr / gtol).max()))
    rest = ~moving & (wc["distance"] > 0)
    assert rest.sum() > 1000
    np.testing.assert_allclose(d[rest], wc["distance"][rest], rtol=1e-4, atol=2e-5 * w)  # axis-aligned grid: no 1x / 2x / 3x split
    # per trajectory the scene cost agrees to the 1e-5 of the fused comparison
    np.testing.assert_allclose(d.sum((1, 2)), wc["distance"].sum((1, 2)), rtol=1e-5, atol=1e-7 * w)
