"""Replay of the REFERENCE's own scene-collision test scenarios -- inputs transcribed verbatim, expectations
as asserted there -- through the CPU oracle (no GPU) and through the HIP kernels (-m gpu).

Sources (all under /root/reference/curobo/tests/_src/):
  geom/sdf/test_voxel_collision.py      ESDF builders :381-440 (restated below in numpy), kernel-level
                                        scenarios :636-1152 (static + swept sphere-voxel collision)
  cost/test_cost_scene_collision.py     table cuboid fixture :54-63 (pose [0.5, 0, 0.3], dims [0.6, 1.0, 0.05]),
                                        empty scene -> all zeros :265-296
The Warp kernels those tests launch cannot run here (no warp, no CUDA), so the reference's asserted
expectations are the pin: every `== pytest.approx(0.0, abs=1e-5)`, `== 0.0`, `> 0.0` and ordering assertion is
checked on both implementations, and the two implementations are compared with each other tightly.
"""

import numpy as np
import pytest

from conftest import load_model

REF = "tests/_src/geom/sdf/test_voxel_collision.py"


# ---------------------------------------------------------------- the reference's ESDF builders, in numpy
def make_empty_esdf(dims=(0.5, 0.5, 0.5), voxel_size=0.02, center=(0.0, 0.0, 0.0), fill_value=1.0):
    """_make_empty_esdf (:381-400): all-free-space grid, every voxel = fill_value (fp16)"""
    n = [int(round(d / voxel_size)) for d in dims]
    return _grid(np.full(n, fill_value, np.float16), n, voxel_size, center)


def make_box_esdf(grid_dims=(0.5, 0.5, 0.5), voxel_size=0.02, grid_center=(0.0, 0.0, 0.0), box_center=(0.0, 0.0, 0.0),
                  box_half=(0.05, 0.05, 0.05)):
    """_make_box_esdf (:403-440): exact box SDF sampled at voxel centres (i - (n - 1) / 2) * voxel_size, fp16"""
    n = [int(round(d / voxel_size)) for d in grid_dims]
    ax = [np.float32(grid_center[k]) + (np.arange(n[k], dtype=np.float32) - np.float32((n[k] - 1) / 2.0)) * np.float32(voxel_size)
          for k in range(3)]
    gx, gy, gz = np.meshgrid(*ax, indexing="ij")
    d = [np.abs(g - np.float32(box_center[k])) - np.float32(box_half[k]) for k, g in enumerate((gx, gy, gz))]
    outside = np.sqrt(sum(np.maximum(x, 0) ** 2 for x in d))
    inside = np.minimum(np.maximum(np.maximum(d[0], d[1]), d[2]), 0)
    return _grid((outside + inside).astype(np.float16), n, voxel_size, grid_center)


def _grid(feat, n, voxel_size, center):
    from curobo_amd.scene import inverse_pose7

    inv = np.zeros((1, 1, 8), np.float32)
    inv[0, 0, :7] = inverse_pose7([*center, 1, 0, 0, 0])
    return {"voxel_params": np.array([[[n[0], n[1], n[2], voxel_size]]], np.float32), "voxel_inv_pose": inv,
            "voxel_enable": np.ones((1, 1), np.uint8), "voxel_count": np.ones((1,), np.int32),
            "voxel_features": feat.reshape(1, 1, -1), "voxel_max_distance": 1000.0}


BOX = dict(grid_dims=(0.5, 0.5, 0.5), voxel_size=0.01, grid_center=(0.0, 0.0, 0.0), box_center=(0.0, 0.0, 0.0), box_half=(0.05, 0.05, 0.05))
R01 = 0.01


def s(*pts, r=R01):
    """[horizon][num_spheres] list of xyz -> (1, H, S, 4)"""
    return np.array([[[[*p, r] for p in row] for row in pts]], np.float32)


# name, reference line, grid, spheres (b, h, S, 4), activation, swept, checks on dist[b, h, s] (and grad)
# check = (kind, index or None): zero -> |x| <= 1e-5 (the reference's approx(0, abs=1e-5) / == 0.0), pos -> x > 0
SCENARIOS = [
    ("sphere_in_free_space_zero_cost", 640, make_empty_esdf((1.0, 1.0, 1.0)), s([(0, 0, 0)]), 0.02, False, [("zero", None)]),
    ("multiple_spheres_in_free_space", 655, make_empty_esdf((1.0, 1.0, 1.0)), s([(0.1, 0, 0)], [(-0.1, 0, 0)], [(0, 0.1, 0)]), 0.02, False,
     [("exact_zero", None)]),
    ("sphere_outside_grid_zero_cost", 674, make_empty_esdf((0.2, 0.2, 0.2)), s([(5.0, 5.0, 5.0)]), 0.02, False, [("zero", None)]),
    ("sphere_inside_box_has_cost", 690, make_box_esdf(**BOX), s([(0, 0, 0)]), 0.02, False, [("pos", None)]),
    ("sphere_far_from_box_zero_cost", 708, make_box_esdf(**BOX), s([(0.2, 0.2, 0.2)]), 0.02, False, [("zero", None)]),
    ("sphere_near_surface_has_cost", 726, make_box_esdf(**BOX), s([(0.05, 0, 0)], r=0.02), 0.02, False, [("pos", None)]),
    ("gradient_nonzero_at_surface", 745, make_box_esdf(**BOX), s([(0.02, 0, 0)]), 0.02, False, [("grad_nonzero", (0, 0, 0))]),
    ("batch_of_spheres", 795, make_box_esdf(**BOX),
     np.array([[[[0.0, 0.0, 0.0, 0.01], [0.2, 0.0, 0.0, 0.01], [0.04, 0.0, 0.0, 0.01]]],
               [[[0.2, 0.2, 0.2, 0.01], [0.0, 0.0, 0.0, 0.01], [-0.2, 0.0, 0.0, 0.01]]]], np.float32), 0.02, False,
     [("pos", (0, 0, 0)), ("zero", (0, 0, 1)), ("zero", (1, 0, 0)), ("pos", (1, 0, 1))]),
    ("horizon_dimension", 821, make_box_esdf(**BOX), s([(0, 0, 0)], [(0.03, 0, 0)], [(0.06, 0, 0)], [(0.2, 0, 0)]), 0.02, False,
     [("pos", (0, 0, 0)), ("zero", (0, 3, 0))]),
    ("swept_horizon1_no_sweep", 882, make_empty_esdf((1.0, 1.0, 1.0)), s([(0, 0, 0)]), 0.02, True, [("zero", None)]),
    ("swept_horizon2_minimal_sweep", 895, make_empty_esdf((1.0, 1.0, 1.0)), s([(-0.1, 0, 0)], [(0.1, 0, 0)]), 0.02, True, [("exact_zero", None)]),
    ("swept_free_space_zero_cost", 912, make_empty_esdf((1.0, 1.0, 1.0)), s([(-0.3, 0, 0)], [(-0.1, 0, 0)], [(0.1, 0, 0)], [(0.3, 0, 0)]),
     0.02, True, [("exact_zero", None)]),
    ("swept_outside_grid_zero_cost", 931, make_empty_esdf((0.2, 0.2, 0.2)), s([(5.0, 0, 0)], [(5.1, 0, 0)]), 0.02, True, [("exact_zero", None)]),
    ("swept_through_box_has_cost", 951, make_box_esdf(**BOX), s([(-0.2, 0, 0)], [(0, 0, 0)], [(0, 0, 0)], [(0.2, 0, 0)]), 0.02, True,
     [("pos", (0, 1, 0)), ("pos", (0, 2, 0))]),
    ("swept_detects_intermediate_collision", 978, make_box_esdf(**BOX), s([(-0.1, 0, 0)], [(0.1, 0, 0)]), 0.02, True, [("sum_pos", None)]),
    ("static_endpoints_of_the_above_are_free", 1002, make_box_esdf(**BOX), s([(-0.1, 0, 0)], [(0.1, 0, 0)]), 0.02, False, [("zero", None)]),
    ("swept_stationary_matches_static", 1014, make_box_esdf(**BOX), s([(0, 0, 0)], [(0, 0, 0)], [(0, 0, 0)]), 0.02, True, [("all_pos", None)]),
    ("static_of_the_above", 1033, make_box_esdf(**BOX), s([(0, 0, 0)], [(0, 0, 0)], [(0, 0, 0)]), 0.02, False, [("all_pos", None)]),
    ("swept_gradient_nonzero_on_collision", 1043, make_box_esdf(**BOX), s([(-0.1, 0, 0)], [(0.02, 0, 0)], [(0.1, 0, 0)]), 0.02, True,
     [("grad_nonzero", (0, 1, 0))]),
    ("swept_far_from_box_zero_cost", 1067, make_box_esdf(**BOX), s([(0.15, 0.15, 0.15)], [(0.16, 0.15, 0.15)], [(0.17, 0.15, 0.15)]), 0.02, True,
     [("exact_zero", None)]),
    ("batch_swept_collision", 1092, make_box_esdf(**BOX),
     np.array([[[[-0.1, 0, 0, 0.01]], [[0.0, 0, 0, 0.01]], [[0.1, 0, 0, 0.01]]],
               [[[0.2, 0.2, 0.0, 0.01]], [[0.2, 0.2, 0.01, 0.01]], [[0.2, 0.2, 0.02, 0.01]]]], np.float32), 0.02, True,
     [("sum_pos", (0,)), ("sum_zero", (1,))]),
    ("multi_sphere_swept", 1126, make_box_esdf(**BOX),
     np.array([[[[-0.1, 0, 0, 0.01], [0.2, 0.2, 0.0, 0.01]], [[0.0, 0, 0, 0.01], [0.2, 0.2, 0.01, 0.01]],
                [[0.1, 0, 0, 0.01], [0.2, 0.2, 0.02, 0.01]]]], np.float32), 0.02, True,
     [("pos", (0, 1, 0)), ("sum_zero", (0, slice(None), 1))]),
]


def _check(name, dist, grad, checks):
    for kind, idx in checks:
        x = dist if idx is None else dist[idx]
        if kind == "zero":
            assert np.all(np.abs(x) <= 1e-5), (name, kind, x)
        elif kind == "exact_zero":
            assert np.all(x == 0.0), (name, kind, x)
        elif kind in ("pos", "all_pos"):
            assert np.all(x > 0.0), (name, kind, x)
        elif kind == "sum_pos":
            assert float(np.sum(x)) > 0.0, (name, kind, x)
        elif kind == "sum_zero":
            assert abs(float(np.sum(x))) <= 1e-5, (name, kind, x)
        elif kind == "grad_nonzero":
            assert float(np.abs(grad[idx][:3]).sum()) > 0.0, (name, kind, grad[idx])
        else:
            raise AssertionError(kind)


def _oracle_run(oracle, grid, spheres, eta, swept):
    r = oracle.scene_collision(spheres, grid, 1.0, eta, sweep=swept)
    return r["distance"], r["gradient"]


def _hip_run(device, grid, spheres, eta, swept):
    import torch

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    scene = SceneData.from_arrays(grid, device)
    b, h, S, _ = spheres.shape
    dist = torch.full((b, h, S), 7.0, device=device)  # the kernel must overwrite every entry
    grad = torch.full((b, h, S, 4), 7.0, device=device)
    Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(spheres, device=device), scene.struct, torch.tensor([1.0], device=device),
                                 torch.tensor([eta], device=device), None, b, h, S, False, 3 if swept else 0, False, None)
    torch.cuda.synchronize()
    return dist.cpu().numpy(), grad.cpu().numpy()


@pytest.mark.parametrize("scn", SCENARIOS, ids=[f"{s_[0]}@{s_[1]}" for s_ in SCENARIOS])
def test_reference_voxel_scenarios_oracle(scn, oracle):
    name, _line, grid, spheres, eta, swept, checks = scn
    dist, grad = _oracle_run(oracle, grid, spheres, eta, swept)
    assert dist.shape == spheres.shape[:3]
    _check(name, dist, grad, checks)


def test_reference_inside_deeper_has_higher_cost_oracle(oracle):
    """:764-790"""
    grid = make_box_esdf(**BOX)
    deep, _ = _oracle_run(oracle, grid, s([(0, 0, 0)]), 0.02, False)
    edge, _ = _oracle_run(oracle, grid, s([(0.04, 0, 0)]), 0.02, False)
    assert deep.item() > edge.item() > 0.0


def test_reference_update_features_inplace_oracle(oracle):
    """:851-873: the same store, features replaced in place, collision appears"""
    grid = make_empty_esdf((0.5, 0.5, 0.5), 0.01)
    before, _ = _oracle_run(oracle, grid, s([(0, 0, 0)]), 0.02, False)
    grid["voxel_features"][...] = make_box_esdf(**BOX)["voxel_features"]
    after, _ = _oracle_run(oracle, grid, s([(0, 0, 0)]), 0.02, False)
    assert abs(before.item()) <= 1e-5 and after.item() > 0.0


def test_reference_surface_penetration_value(oracle):
    """:737-739 states the expected penetration for the sphere touching the box face: -sdf + (r + eta) =
    0.04; the activation (wp_collision_common.py:11-38) is quadratic below eta and linear above:
    cost = 0.04 - eta / 2 = 0.03 (weight 1).  The fp16 grid + trilinear lookup reproduce sdf = 0 at the
    face to within half a voxel's fp16 rounding."""
    dist, _ = _oracle_run(oracle, make_box_esdf(**BOX), s([(0.05, 0, 0)], r=0.02), 0.02, False)
    assert dist.item() == pytest.approx(0.03, abs=1.5e-3)


# ---------------------------------------------------------------- cuboid fixtures of the cost test
TABLE = [[{"dims": [0.6, 1.0, 0.05], "pose": [0.5, 0.0, 0.3, 1.0, 0.0, 0.0, 0.0]}]]  # test_cost_scene_collision.py:54-63


def _franka_spheres(oracle, q):
    model = load_model("franka")
    return oracle.kinematics_forward(np.asarray(q, np.float32).reshape(-1, model.num_dof), model.as_dict())["robot_spheres"][None]


def test_reference_table_fixture_and_empty_scene_oracle(oracle):
    """cost/test_cost_scene_collision.py: the table cuboid in front of the Franka (:54-63) and the empty
    scene (:265-296, `assert torch.all(result == 0)`).  Known geometry: at the default joint position
    (franka.yml) the arm is above / behind the table -> zero cost; reaching down into the table -> hits
    on the forearm / hand spheres only, and exactly the spheres whose centre-to-slab distance is below r + eta."""
    from curobo_amd.scene import cuboid_scene_arrays

    model = load_model("franka")
    q_default = np.asarray(model.cspace["default_joint_position"], np.float32)
    sph = _franka_spheres(oracle, q_default)
    arrays = cuboid_scene_arrays(TABLE)
    r = oracle.scene_collision(sph, arrays, 1.0, 0.0)
    empty = cuboid_scene_arrays([[{"dims": [0.1, 0.1, 0.1], "pose": [0, 0, 0, 1, 0, 0, 0], "enable": False}]])
    assert np.all(oracle.scene_collision(sph, empty, 1.0, 0.0)["distance"] == 0.0)
    # independent closed form for the axis-aligned slab: hit <=> exact box SDF of the centre < r
    c, rad = sph[0, 0, :, :3].astype(np.float64), sph[0, 0, :, 3].astype(np.float64)
    qd = np.abs(c - np.array([0.5, 0.0, 0.3])) - 0.5 * np.array([0.6, 1.0, 0.05])
    sdf = np.linalg.norm(np.maximum(qd, 0), axis=-1) + np.minimum(qd.max(-1), 0)
    want = (sdf < rad) & (rad >= 0)
    assert np.array_equal(r["distance"][0, 0] > 0, want)
    q_reach = np.array([0.0, 0.9, 0.0, -1.2, 0.0, 2.2, 0.8], np.float32)  # hand pushed through the table top
    sph2 = _franka_spheres(oracle, q_reach)
    r2 = oracle.scene_collision(sph2, arrays, 1.0, 0.0)
    c, rad = sph2[0, 0, :, :3].astype(np.float64), sph2[0, 0, :, 3].astype(np.float64)
    qd = np.abs(c - np.array([0.5, 0.0, 0.3])) - 0.5 * np.array([0.6, 1.0, 0.05])
    sdf = np.linalg.norm(np.maximum(qd, 0), axis=-1) + np.minimum(qd.max(-1), 0)
    want = (sdf < rad - 1e-6) & (rad >= 0)
    got = r2["distance"][0, 0] > 0
    assert want.sum() >= 3 and np.array_equal(got | (np.abs(sdf - rad) < 1e-5), want | (np.abs(sdf - rad) < 1e-5))
    # cost value (weight 1, eta 0): penetration depth r - sdf for spheres outside / crossing the faces
    outside = want & (sdf > 0)
    np.testing.assert_allclose(r2["distance"][0, 0][outside], (rad - sdf)[outside], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------- the same scenarios through the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("scn", SCENARIOS, ids=[f"{s_[0]}@{s_[1]}" for s_ in SCENARIOS])
def test_reference_voxel_scenarios_hip(scn, oracle, device):
    name, _line, grid, spheres, eta, swept, checks = scn
    dist, grad = _hip_run(device, grid, spheres, eta, swept)
    _check(name, dist, grad, checks)
    rd, rg = _oracle_run(oracle, grid, spheres, eta, swept)
    assert np.array_equal(dist > 0, rd > 0), "bit-exact collision-hit indices"
    np.testing.assert_allclose(dist, rd, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(grad, rg, atol=2e-5, rtol=1e-4)


@pytest.mark.gpu
def test_reference_table_fixture_hip(oracle, device):
    import torch

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, cuboid_scene_arrays

    arrays = cuboid_scene_arrays(TABLE)
    q = np.stack([np.asarray(load_model("franka").cspace["default_joint_position"], np.float32),
                  np.array([0.0, 0.9, 0.0, -1.2, 0.0, 2.2, 0.8], np.float32)])
    sph = np.concatenate([_franka_spheres(oracle, q[0]), _franka_spheres(oracle, q[1])], axis=0)  # (2, 1, S, 4)
    ref = oracle.scene_collision(sph, arrays, 1.0, 0.0)
    scene = SceneData.from_arrays(arrays, device)
    b, h, S, _ = sph.shape
    dist, grad = torch.zeros(b, h, S, device=device), torch.zeros(b, h, S, 4, device=device)
    Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
                                 torch.tensor([0.0], device=device), None, b, h, S, False, 0, False, None)
    torch.cuda.synchronize()
    assert np.array_equal(dist.cpu().numpy() > 0, ref["distance"] > 0) and (ref["distance"][1] > 0).sum() >= 3
    np.testing.assert_allclose(dist.cpu().numpy(), ref["distance"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), ref["gradient"], atol=1e-5, rtol=1e-4)
