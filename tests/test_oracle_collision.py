"""Pins for the collision oracles: analytic SDF expectations, finite differences, the
reference's property tests (curobo/tests/_src/cost/test_cost_self_collision.py,
test_cost_scene_collision.py, geom/sdf/test_voxel_collision.py:636-1437)."""

import numpy as np
import pytest

from conftest import sample_q
from curobo_amd.scene import cuboid_scene_arrays, voxel_grid_from_sdf


def _spheres(pts, r=0.05):
    pts = np.asarray(pts, np.float32).reshape(1, 1, -1, 3)
    return np.concatenate([pts, np.full(pts.shape[:-1] + (1,), r, np.float32)], -1)


TABLE = [{"dims": [0.6, 1.0, 0.05], "pose": [0.5, 0, 0.3, 1, 0, 0, 0]}]  # reference test_cost_scene_collision.py:54-77


def test_cuboid_free_space_is_zero_and_inside_positive(oracle):
    arr = cuboid_scene_arrays([TABLE])
    out = oracle.scene_collision(_spheres([[0.5, 0, 1.0], [0.5, 0, 0.3], [0.5, 0, 0.34]]), arr, 1.0, 0.02)
    d = out["distance"][0, 0]
    assert d[0] == 0.0 and d[1] > 0 and d[2] > 0 and d[1] > d[2], d  # deeper => higher
    assert (out["gradient"][0, 0, 0] == 0).all()


def test_cuboid_cost_value_linear_and_quadratic_regions(oracle):
    """activation (wp_collision_common.py:11-38): pen > eta -> pen - eta/2 ; else pen^2 / (2 eta)"""
    arr = cuboid_scene_arrays([TABLE])
    eta, r, w = 0.02, 0.05, 3.0
    top = 0.3 + 0.025
    # sphere centre 0.06 above the top face: sdf = 0.06, pen = r + eta - sdf = 0.01 (quadratic region)
    # sphere centre 0.02 above: pen = 0.05 (linear region)
    out = oracle.scene_collision(_spheres([[0.5, 0, top + 0.06], [0.5, 0, top + 0.02]], r), arr, w, eta)
    np.testing.assert_allclose(out["distance"][0, 0], [w * 0.5 * 0.01 ** 2 / eta, w * (0.05 - 0.5 * eta)], rtol=1e-4)
    # gradient = -w * gs * d(sdf)/dx: pushing up (+z) reduces the cost
    np.testing.assert_allclose(out["gradient"][0, 0, :, :3], [[0, 0, -w * 0.01 / eta], [0, 0, -w]], atol=1e-4)


def test_cuboid_gradient_matches_finite_differences(oracle):
    c, s = np.cos(0.35), np.sin(0.35)
    arr = cuboid_scene_arrays([[{"dims": [0.4, 0.3, 0.2], "pose": [0.1, -0.2, 0.3, c, s * 0.6, 0, s * 0.8]},
                                {"dims": [0.2, 0.5, 0.3], "pose": [0.3, 0.1, 0.2, 1, 0, 0, 0]}]])
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.1, 0.6, size=(400, 3)).astype(np.float32)
    sp = _spheres(pts, 0.06)
    out = oracle.scene_collision(sp, arr, 2.0, 0.03)
    eps = 2e-4
    checked = 0
    for k in np.nonzero(out["distance"][0, 0] > 1e-4)[0][:60]:
        fd = np.zeros(3)
        for a in range(3):
            p1, p0 = sp.copy(), sp.copy()
            p1[0, 0, k, a] += eps
            p0[0, 0, k, a] -= eps
            fd[a] = (float(oracle.scene_collision(p1, arr, 2.0, 0.03)["distance"][0, 0, k])
                     - float(oracle.scene_collision(p0, arr, 2.0, 0.03)["distance"][0, 0, k])) / (2 * eps)
        g = out["gradient"][0, 0, k, :3]
        if np.linalg.norm(fd - g) < 0.05 * max(1.0, np.linalg.norm(fd)):
            checked += 1
    assert checked >= 50  # a few points straddle SDF kinks (edges, inside/outside switch)


def test_disabled_and_uncounted_obstacles_are_ignored(oracle):
    obs = [dict(TABLE[0]), {"dims": [0.2, 0.2, 0.2], "pose": [0.5, 0, 1.0, 1, 0, 0, 0], "enable": False}]
    arr = cuboid_scene_arrays([obs])
    sp = _spheres([[0.5, 0, 1.0]])
    assert oracle.scene_collision(sp, arr, 1.0, 0.02)["distance"][0, 0, 0] == 0.0
    arr["cuboid_enable"][0, 1] = 1
    assert oracle.scene_collision(sp, arr, 1.0, 0.02)["distance"][0, 0, 0] > 0.0
    arr["cuboid_count"][0] = 1  # beyond count -> ignored even if enabled
    assert oracle.scene_collision(sp, arr, 1.0, 0.02)["distance"][0, 0, 0] == 0.0


def test_negative_radius_spheres_are_skipped(oracle):
    arr = cuboid_scene_arrays([TABLE])
    sp = _spheres([[0.5, 0, 0.3]], r=-100.0)
    out = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=True)
    assert out["distance"][0, 0, 0] == 0.0


def test_swept_equals_static_when_stationary(oracle):
    """reference test_voxel_collision.py: swept == static when the trajectory does not move"""
    arr = cuboid_scene_arrays([TABLE])
    sp = np.repeat(_spheres([[0.5, 0, 0.33], [0.2, 0.1, 0.31], [0.0, 0.0, 1.0]]), 5, axis=1)
    a = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=False)
    b = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=True)
    np.testing.assert_array_equal(a["distance"], b["distance"])
    np.testing.assert_array_equal(a["gradient"], b["gradient"])


def test_sweep_catches_tunnelling(oracle):
    """a sphere that jumps across a thin wall between two steps is only seen by the sweep"""
    wall = cuboid_scene_arrays([[{"dims": [0.02, 1.0, 1.0], "pose": [0.5, 0, 0.5, 1, 0, 0, 0]}]])
    sp = np.zeros((1, 3, 1, 4), np.float32)
    sp[0, :, 0] = [[0.35, 0, 0.5, 0.03], [0.42, 0, 0.5, 0.03], [0.62, 0, 0.5, 0.03]]
    st = oracle.scene_collision(sp, wall, 1.0, 0.01, sweep=False)["distance"][0, :, 0]
    sw = oracle.scene_collision(sp, wall, 1.0, 0.01, sweep=True)["distance"][0, :, 0]
    assert (st == 0).all() and sw[1] > 0


def test_speed_metric_scales_by_velocity(oracle):
    arr = cuboid_scene_arrays([TABLE])
    z = 0.3 + 0.025 + 0.03
    sp = np.zeros((1, 3, 1, 4), np.float32)
    sp[0, :, 0] = [[0.40, 0, z, 0.05], [0.50, 0, z, 0.05], [0.60, 0, z, 0.05]]
    a = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=True, enable_speed_metric=False)
    b = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=True, enable_speed_metric=True, speed_dt=0.1)
    # straight line at 1 m/s: interior point cost scales by |v| = 1, end points untouched
    np.testing.assert_allclose(b["distance"][0, 1, 0], a["distance"][0, 1, 0] * 1.0, rtol=1e-5)
    b2 = oracle.scene_collision(sp, arr, 1.0, 0.02, sweep=True, enable_speed_metric=True, speed_dt=0.05)
    np.testing.assert_allclose(b2["distance"][0, 1, 0], a["distance"][0, 1, 0] * 2.0, rtol=1e-5)
    np.testing.assert_array_equal(b2["distance"][0, [0, 2], 0], a["distance"][0, [0, 2], 0])
    # gradient component along the motion direction is projected out (I - v v^T)
    assert abs(b2["gradient"][0, 1, 0, 0]) < 1e-5


def _box_sdf(p, c=(0.0, 0.0, 0.0), h=(0.2, 0.15, 0.1)):
    q = np.abs(p - np.asarray(c)) - np.asarray(h)
    return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)


def test_voxel_matches_analytic_box(oracle):
    """reference test_voxel_collision.py analytic-box suite: free space zero, inside positive,
    deeper => higher, and values close to the analytic cuboid cost."""
    vox = voxel_grid_from_sdf(_box_sdf, (64, 64, 64), 0.02, max_distance=100.0)
    cub = cuboid_scene_arrays([[{"dims": [0.4, 0.3, 0.2], "pose": [0, 0, 0, 1, 0, 0, 0]}]])
    rng = np.random.default_rng(5)
    pts = rng.uniform(-0.5, 0.5, size=(500, 3)).astype(np.float32)
    sp = _spheres(pts, 0.04)
    a = oracle.scene_collision(sp, vox, 1.0, 0.02)
    b = oracle.scene_collision(sp, cub, 1.0, 0.02)
    np.testing.assert_allclose(a["distance"], b["distance"], atol=6e-3)  # trilinear + fp16 error
    hit = b["distance"][0, 0] > 5e-3
    cosang = (a["gradient"][0, 0, hit, :3] * b["gradient"][0, 0, hit, :3]).sum(-1) / (
        np.linalg.norm(a["gradient"][0, 0, hit, :3], axis=-1) * np.linalg.norm(b["gradient"][0, 0, hit, :3], axis=-1) + 1e-9)
    assert np.median(cosang) > 0.98
    deep = oracle.scene_collision(_spheres([[0, 0, 0.0], [0, 0, 0.08], [0, 0, 0.3]], 0.04), vox, 1.0, 0.02)["distance"][0, 0]
    assert deep[0] > deep[1] > 0 and deep[2] == 0


def test_voxel_outside_grid_is_free(oracle):
    vox = voxel_grid_from_sdf(_box_sdf, (16, 16, 16), 0.05, max_distance=100.0)
    out = oracle.scene_collision(_spheres([[3.0, 0, 0], [0.41, 0, 0]], 0.1), vox, 1.0, 0.02, sweep=True)
    assert out["distance"][0, 0, 0] == 0.0  # far outside: default value = max_dist => no collision


def test_self_collision_basics(oracle, franka):
    """reference test_cost_self_collision.py:146-200: default configuration is collision free,
    a folded arm is not; gradient only on the arg-max pair; padding inflates."""
    md = franka.as_dict()
    q_default = np.array([[0.0, -1.3, 0.0, -2.5, 0.0, 1.5, 0.8]], np.float32)
    sph = oracle.kinematics_forward(q_default, md)["robot_spheres"]
    r = oracle.self_collision(sph, franka.sphere_padding, franka.collision_pairs, 1.0)
    assert r["distance"][0] == 0.0 and (r["pair_idx"] == -1).all() and not r["gradient"].any()
    q_fold = np.array([[0.0, 1.7, 0.0, -3.0, 0.0, 3.7, 0.0]], np.float32)
    sph = oracle.kinematics_forward(q_fold, md)["robot_spheres"]
    r = oracle.self_collision(sph, franka.sphere_padding, franka.collision_pairs, 2.0, store_pair_distance=True)
    assert r["distance"][0] > 0
    i, j = r["pair_idx"][0]
    assert r["sparse_index"][0].sum() == 2 and r["sparse_index"][0, i] == 1 and r["sparse_index"][0, j] == 1
    nz = np.nonzero(np.abs(r["gradient"][0]).sum(-1))[0]
    assert set(nz) == {i, j}
    np.testing.assert_allclose(r["gradient"][0, i, :3], -r["gradient"][0, j, :3])
    assert r["distance"][0] == pytest.approx(0.5 * 2.0 * r["pair_distance"][0].max(), rel=1e-6)
    # the winning pair index is the FIRST maximal entry (canonical tie rule)
    k = int(np.argmax(r["pair_distance"][0]))
    assert tuple(franka.collision_pairs[k]) == (i, j)


def test_self_collision_tie_rule_lowest_pair_index(oracle):
    sph = np.zeros((1, 4, 4), np.float32)
    sph[0, :, :3] = [[0, 0, 0], [0.1, 0, 0], [0, 0, 1], [0.1, 0, 1]]
    sph[0, :, 3] = 0.1
    pairs = np.array([[2, 3], [0, 1]], np.int16)  # both pairs penetrate by exactly the same amount
    r = oracle.self_collision(sph, np.zeros(4, np.float32), pairs, 1.0)
    assert tuple(r["pair_idx"][0]) == (2, 3)


def test_self_collision_gradient_matches_finite_differences(oracle, franka):
    md = franka.as_dict()
    q = np.array([[0.0, 1.7, 0.0, -3.0, 0.0, 3.7, 0.0]], np.float32)
    sph = oracle.kinematics_forward(q, md)["robot_spheres"]
    r = oracle.self_collision(sph, franka.sphere_padding, franka.collision_pairs, 1.0)
    i = r["pair_idx"][0, 0]
    eps = 1e-4
    for a in range(3):
        p1, p0 = sph.copy(), sph.copy()
        p1[0, i, a] += eps
        p0[0, i, a] -= eps
        fd = (oracle.self_collision(p1, franka.sphere_padding, franka.collision_pairs, 1.0)["distance"][0]
              - oracle.self_collision(p0, franka.sphere_padding, franka.collision_pairs, 1.0)["distance"][0]) / (2 * eps)
        assert fd == pytest.approx(r["gradient"][0, i, a], abs=2e-3)


def test_device_frame_arithmetic_differs_from_the_reference_only_on_resting_spheres(oracle, franka):
    """orc_set_frame_arithmetic(1) (the HIP path's rotation-matrix fma form of the world -> obstacle-frame transform) against
    the reference's quat_rotate form on the same spheres, swept: per sphere the two costs agree to rounding EXCEPT where the
    sphere is stationary up to rounding (its obstacle-frame distance to a neighbour point is zero in one arithmetic and an
    ulp in the other -> one more or one fewer copy of its centre sample, wp_sweep_collision_kernel.py:186-203).  The
    difference there is a whole number of centre-sample costs.  Rotated cuboids, so that the transform rounds at all."""
    from curobo_amd.workloads import seed_knots, start_configuration
    from oracle.rollout_ref import rollout_cost_and_gradient

    rng = np.random.default_rng(3)
    world = []
    for p in ([0.45, 0.0, 0.25], [0.3, 0.4, 0.5], [0.2, -0.4, 0.4], [0.55, 0.2, 0.7]):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        world.append({"dims": [0.35, 0.3, 0.3], "pose": [*p, *[float(v) for v in q]]})
    arr = cuboid_scene_arrays([world])
    knots = seed_knots(franka, 96, 12, seed=2)  # the seeds come to rest at their last knot: stationary spheres
    ref = rollout_cost_and_gradient(oracle, franka.as_dict(), arr, knots, start_configuration(franka))
    sph = ref["robot_spheres"]
    eta, w = 0.02, 1.0
    assert oracle.frame_arithmetic() == "reference"
    a = oracle.scene_collision(sph, arr, w, eta, sweep=True)["distance"]
    centre = oracle.scene_collision(sph, arr, w, eta, sweep=False)["distance"]
    oracle.set_frame_arithmetic("device")
    try:
        b = oracle.scene_collision(sph, arr, w, eta, sweep=True)["distance"]
        centre_b = oracle.scene_collision(sph, arr, w, eta, sweep=False)["distance"]
    finally:
        oracle.set_frame_arithmetic("reference")
    np.testing.assert_allclose(centre_b, centre, rtol=2e-5, atol=1e-7)  # no decision in the discrete kernel
    p = sph[..., :3]
    stepn = np.linalg.norm(np.diff(p, axis=1), axis=-1)
    still = np.zeros(p.shape[:3], bool)
    still[:, 1:] |= stepn < 1e-5
    still[:, :-1] |= stepn < 1e-5
    # (a last bit of a local coordinate is ~6e-8 m; in the quadratic region of the activation a cost of 5e-3 moves by 1e-4 of itself)
    differs = np.abs(a - b) > 5e-4 * np.maximum(np.abs(a), np.abs(b)) + 1e-7
    assert (a > 0).sum() > 1000 and (still & (a > 0)).sum() > 100
    assert not (differs & ~still).any(), "moving spheres see the same sweep in both arithmetics"
    # where they differ: a whole number (1 or 2) of centre samples.  One obstacle in reach per resting sphere here would make
    # this exact; with several, the centre cost is their sum and each may flip on its own: allow 0 .. 2 x the centre cost.
    d = np.abs(a - b)[differs]
    assert (d <= 2.0 * centre[differs] * (1 + 1e-4) + 1e-6).all()
    assert differs.sum() > 0, "the case this mode exists for must occur in the sample"
