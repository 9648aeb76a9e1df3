"""``AttachmentManager`` and the obstacles-by-name members of ``SceneData`` on the host (no kernel runs: the writes are plain
tensor updates).  The cases follow the reference's ``tests/_src/collision/test_attachment_manager.py`` (fit, update, multi
environment, attach / detach round trips, world obstacles switched off and on, attach_from_scene and its errors) on the packaged
Franka (four ``attached_object`` slots); the case with a world pose offset needs FK and is in ``tests/test_gpu_api.py``."""
import numpy as np
import pytest
import torch

from curobo_amd.attachment_manager import AttachmentManager, fit_spheres_to_obstacle
from curobo_amd.kinematics import Kinematics, KinematicsCfg
from curobo_amd.scene.config import scene_from_config
from curobo_amd.scene.data import cuboid_scene_arrays
from curobo_amd.scene.types import Capsule, Cuboid, Cylinder, SceneCfg, Sphere
from curobo_amd.types import JointState, Pose

Q = [0.0, -1.2, 0.0, -2.0, 0.0, 1.0, 0.0]


@pytest.fixture()
def kin():
    return Kinematics(KinematicsCfg.from_packaged("franka", device="cpu"))


@pytest.fixture()
def cube():
    return Cuboid(name="test_cube", pose=[0.5, 0.0, 0.5, 1.0, 0.0, 0.0, 0.0], dims=[0.05, 0.05, 0.05])


def scene_with(*obstacles, cache=10):
    cfg = SceneCfg()
    for o in obstacles:
        cfg.add_obstacle(o)
    return scene_from_config(cfg, "cpu", cache={"cuboid": cache})


def grasp():
    return JointState.from_position(torch.tensor([Q]))


# ---------------------------------------------------------------------------------------------------------------- fit
def test_fit_is_inside_the_obstacle_and_within_the_budget(kin, cube):
    m = AttachmentManager(kin)
    for obstacle, budget in ((cube, 10), (Cuboid("slab", [0.1, 0.2, 0.3, 0.924, 0.0, 0.383, 0.0], dims=[0.3, 0.1, 0.02]), 12),
                             (Cuboid("slab", [0, 0, 0, 1, 0, 0, 0], dims=[0.3, 0.1, 0.02]), None)):
        s = m.fit_spheres([obstacle], num_spheres=budget)
        assert s.dim() == 2 and s.shape[1] == 4 and s.shape[0] > 0 and (s[:, 3] > 0).all()
        assert budget is None or s.shape[0] <= budget
        assert m._last_fit_result.num_spheres == s.shape[0]
        # every sphere lies inside the box: |R^T (c - t)| + r <= half extent
        from curobo_amd.scene.types import Pose7

        P = Pose7(obstacle.pose)
        local = (s[:, :3].double().numpy() - P.t) @ P.R
        assert (np.abs(local) + s[:, 3:4].double().numpy() <= 0.5 * np.asarray(obstacle.dims) + 1e-6).all()
    # without a budget the lattice leaves no gap along the long axes (pitch = radius)
    s = m.fit_spheres([Cuboid("slab", [0, 0, 0, 1, 0, 0, 0], dims=[0.3, 0.1, 0.02])])
    xs = np.unique(np.round(s[:, 0].numpy(), 6))
    assert np.diff(xs).max() <= 2 * float(s[0, 3]) + 1e-6


def test_fit_of_the_analytic_primitives():
    ball = fit_spheres_to_obstacle(Sphere("ball", pose=[0.1, 0.2, 0.3, 1, 0, 0, 0], radius=0.07))
    np.testing.assert_allclose(ball, [[0.1, 0.2, 0.3, 0.07]], atol=1e-7)
    cap = fit_spheres_to_obstacle(Capsule("cap", pose=[0, 0, 1.0, 1, 0, 0, 0], radius=0.05, base=[0, 0, -0.1], tip=[0, 0, 0.1]))
    assert np.allclose(cap[:, 3], 0.05) and np.isclose(cap[:, 2].min(), 0.9) and np.isclose(cap[:, 2].max(), 1.1)
    cyl = fit_spheres_to_obstacle(Cylinder("cyl", pose=[0, 0, 0, 1, 0, 0, 0], radius=0.05, height=0.4), num_spheres=6)
    assert cyl.shape[0] <= 6 and np.allclose(cyl[:, 3], 0.05) and np.abs(cyl[:, 2]).max() <= 0.15 + 1e-6


# ---------------------------------------------------------------------------------------------------------------- update
def test_update_writes_the_link_slots_and_disables_the_rest(kin):
    m, kp = AttachmentManager(kin), kin.kinematics_config
    slots = kp.get_sphere_index_from_link_name("attached_object")
    before = kp.link_spheres.clone()
    sph = torch.zeros(2, 4)
    sph[:, 3] = 0.01
    sph[0, :3] = torch.tensor([0.1, 0.2, 0.3])
    sph[1, 0] = 0.2
    m.update(sph, grasp())
    got = kp.link_spheres[0, slots]
    assert torch.equal(got[:2], sph) and (got[2:, 3] == -100.0).all() and (got[2:, :3] == 0).all()
    other = torch.ones(kp.link_spheres.shape[1], dtype=torch.bool)
    other[slots] = False
    assert torch.equal(kp.link_spheres[0, other], before[0, other])  # the robot's own spheres are untouched
    assert m._attached_link_name == "attached_object"
    with pytest.raises(ValueError):
        m.update(torch.zeros(slots.numel() + 1, 4), grasp())
    m.detach()
    assert torch.equal(kp.link_spheres, before) and m._attached_link_name is None
    m.detach()  # (nothing attached: no effect)


def test_two_environments_and_the_round_trip(kin, cube):
    kp = kin.kinematics_config
    kp.link_spheres = kp.link_spheres.repeat(2, 1, 1)
    kp.reference_link_spheres = kp.reference_link_spheres.repeat(2, 1, 1)
    m = AttachmentManager(kin)
    before = kp.link_spheres.clone()
    q = torch.tensor([Q, [0.5, -0.8, 0.3, -1.5, 0.2, 0.8, 0.1]])
    m.attach(JointState.from_position(q), [cube], num_spheres=3)
    slots = kp.get_sphere_index_from_link_name("attached_object")
    n = m._last_fit_result.num_spheres
    assert (kp.link_spheres[:, slots[:n], 3] > 0).all() and torch.equal(kp.link_spheres[0, slots], kp.link_spheres[1, slots])
    m.detach()
    assert torch.equal(kp.link_spheres, before)
    with pytest.raises(ValueError):
        m.update(torch.zeros(1, 4), JointState.from_position(torch.zeros(3, 7)))  # three states, two sphere sets


# ---------------------------------------------------------------------------------------------------------------- the world's copy
def test_attach_switches_world_obstacles_off_and_detach_back_on(kin, cube):
    scene = scene_with(Cuboid("world_cube", [0.5, 0, 0.5, 1, 0, 0, 0], dims=[0.1, 0.1, 0.1]), Cuboid("table", [0, 0, -0.1, 1, 0, 0, 0], dims=[2, 2, 0.2]))
    m = AttachmentManager(kin, scene)
    _, i = scene.find_obstacle("world_cube")
    assert int(scene.tensors["cuboid_enable"][0, i]) == 1
    m.attach(grasp(), [cube], num_spheres=4, disable_obstacle_names=["world_cube"])
    assert int(scene.tensors["cuboid_enable"][0, i]) == 0 and int(scene.arrays["cuboid_enable"][0, i]) == 0
    assert int(scene.tensors["cuboid_enable"][0, scene.find_obstacle("table")[1]]) == 1
    m.detach()
    assert int(scene.tensors["cuboid_enable"][0, i]) == 1


def test_attach_from_scene(kin):
    scene = scene_with(Cuboid("scene_cube", [0.5, 0, 0.5, 1, 0, 0, 0], dims=[0.05, 0.05, 0.05]))
    m, kp = AttachmentManager(kin, scene), kin.kinematics_config
    m.attach_from_scene(grasp(), ["scene_cube"], num_spheres=4)
    slots = kp.get_sphere_index_from_link_name("attached_object")
    assert (kp.link_spheres[0, slots[: m._last_fit_result.num_spheres], 3] > 0).all()
    assert int(scene.tensors["cuboid_enable"][0, 0]) == 0
    m.detach()
    assert int(scene.tensors["cuboid_enable"][0, 0]) == 1
    with pytest.raises(ValueError):
        m.attach_from_scene(grasp(), ["nonexistent_obstacle"])
    with pytest.raises(ValueError):
        AttachmentManager(kin).attach_from_scene(grasp(), ["scene_cube"])


# ---------------------------------------------------------------------------------------------------------------- obstacles by name
def test_scene_obstacles_by_name_are_in_place_edits_of_the_stores():
    ball = Sphere("ball", pose=[1, 1, 1, 1, 0, 0, 0], radius=0.1)
    box = Cuboid("box", [0.5, 0, 0.5, 1, 0, 0, 0], dims=[0.1, 0.2, 0.3])
    scene = scene_with(box, ball, cache=4)
    ptrs = {k: v.data_ptr() for k, v in scene.tensors.items()}
    assert scene.get_obstacle_names() == ["box", "ball"] and scene.check_obstacle_exists("ball") and not scene.check_obstacle_exists("nope")
    assert scene.find_obstacle("ball") == ("cuboid", 1)
    # pose: what a fresh load of the moved obstacle holds
    new_pose = [0.2, -0.3, 0.4, 0.924, 0.0, 0.383, 0.0]
    scene.update_obstacle_pose("box", new_pose)
    fresh = cuboid_scene_arrays([[{"dims": box.dims, "pose": new_pose}]])
    np.testing.assert_array_equal(scene.tensors["cuboid_inv_pose"][0, 0].numpy(), fresh["cuboid_inv_pose"][0, 0])
    np.testing.assert_array_equal(scene.arrays["cuboid_inv_pose"][0, 0], fresh["cuboid_inv_pose"][0, 0])
    scene.update_obstacle_pose("ball", Pose(torch.tensor([[2.0, 0, 0]]), torch.tensor([[1.0, 0, 0, 0]])))
    np.testing.assert_allclose(scene.tensors["cuboid_inv_pose"][0, 1, :7].numpy(), [-2, 0, 0, 1, 0, 0, 0], atol=1e-7)
    scene.update_obstacle_dims("box", [0.4, 0.5, 0.6])
    np.testing.assert_allclose(scene.tensors["cuboid_dims"][0, 0].numpy(), [0.4, 0.5, 0.6, 0.0])
    with pytest.raises(ValueError):
        scene.update_obstacle_dims("ball", [1, 1, 1])
    scene.enable_obstacle("ball", False)
    assert scene.tensors["cuboid_enable"][0].tolist() == [1, 0, 0, 0]
    scene.enable_obstacle("ball", True)
    # add: the next free slot, as a fresh load would fill it
    cyl = Cylinder("cyl", pose=[0, 0, 1, 1, 0, 0, 0], radius=0.1, height=0.3)
    assert scene.add_obstacle(cyl) == 2 and scene.get_obstacle_names() == ["box", "ball", "cyl"]
    fresh = cuboid_scene_arrays([[{"type": "cylinder", "radius": 0.1, "height": 0.3, "pose": cyl.pose}]])
    np.testing.assert_array_equal(scene.tensors["cuboid_dims"][0, 2].numpy(), fresh["cuboid_dims"][0, 0])
    np.testing.assert_array_equal(scene.tensors["cuboid_inv_pose"][0, 2].numpy(), fresh["cuboid_inv_pose"][0, 0])
    assert int(scene.tensors["cuboid_count"][0]) == 3 and int(scene.tensors["cuboid_enable"][0, 2]) == 1
    with pytest.raises(RuntimeError):
        scene.add_obstacle(cyl)  # the name is taken
    scene.add_obstacle(Cuboid("last", [0, 0, 2, 1, 0, 0, 0], dims=[0.1, 0.1, 0.1]))
    with pytest.raises(RuntimeError):
        scene.add_obstacle(Cuboid("one too many", [0, 0, 3, 1, 0, 0, 0], dims=[0.1, 0.1, 0.1]))
    with pytest.raises(ValueError):
        scene.enable_obstacle("nope")
    assert {k: v.data_ptr() for k, v in scene.tensors.items()} == ptrs  # (captured graphs keep reading the same memory)
    with pytest.raises(ValueError):
        scene_from_config(SceneCfg(cuboid=[box, Cuboid("b2", box.pose, dims=box.dims)]), "cpu", cache={"cuboid": 1})


def test_clear_forgets_the_obstacles_and_frees_their_slots():
    """``SceneData.clear`` as the reference's stores clear (data_cuboid.py:411-424): enable AND count to zero, names
    forgotten -- the same world can be added again, a cleared obstacle cannot be switched back on."""
    box = Cuboid("box", [0.5, 0, 0.5, 1, 0, 0, 0], dims=[0.1, 0.2, 0.3])
    ball = Sphere("ball", pose=[1, 1, 1, 1, 0, 0, 0], radius=0.1)
    scene = scene_with(box, ball, cache=2)
    ptrs = {k: v.data_ptr() for k, v in scene.tensors.items()}
    scene.clear()
    assert scene.get_obstacle_names() == [] and not scene.check_obstacle_exists("box")
    assert int(scene.tensors["cuboid_count"][0]) == 0 and scene.tensors["cuboid_enable"][0].tolist() == [0, 0]
    assert int(scene.arrays["cuboid_count"][0]) == 0
    with pytest.raises(ValueError):
        scene.enable_obstacle("box", True)  # gone, not merely switched off
    # the same world again: the full store has room, the names are free
    assert scene.add_obstacle(box) == 0 and scene.add_obstacle(ball) == 1
    assert scene.get_obstacle_names() == ["box", "ball"] and scene.tensors["cuboid_enable"][0].tolist() == [1, 1]
    assert {k: v.data_ptr() for k, v in scene.tensors.items()} == ptrs
    # one environment of two
    cfg = SceneCfg()
    cfg.add_obstacle(box)
    two = scene_from_config([cfg, cfg], "cpu", cache={"cuboid": 2}) if _takes_env_list() else None
    if two is not None:
        two.clear(env_idx=1)
        assert two.get_obstacle_names(0) == ["box"] and two.get_obstacle_names(1) == []
        assert two.tensors["cuboid_count"].tolist() == [1, 0]


def _takes_env_list():
    try:
        cfg = SceneCfg()
        cfg.add_obstacle(Cuboid("b", [0, 0, 0, 1, 0, 0, 0], dims=[0.1, 0.1, 0.1]))
        return scene_from_config([cfg, cfg], "cpu", cache={"cuboid": 2}).num_envs == 2
    except Exception:
        return False


def test_conservative_fit_covers_the_obstacle():
    """``conservative=True``: every point of the (rotated) box lies in some sphere, within the budget; the default fit stays inside"""
    from curobo_amd.scene.types import Pose7

    rng = np.random.default_rng(3)
    for dims, budget in (([0.3, 0.1, 0.02], 12), ([0.05, 0.05, 0.05], 4), ([0.2, 0.15, 0.1], None), ([0.4, 0.02, 0.02], 3)):
        box = Cuboid("b", [0.1, -0.2, 0.3, 0.924, 0.0, 0.383, 0.0], dims=dims)
        s = fit_spheres_to_obstacle(box, budget, conservative=True)
        assert budget is None or s.shape[0] <= budget
        P = Pose7(box.pose)
        pts = P.transform(rng.uniform(-0.5, 0.5, (4000, 3)) * np.asarray(dims))
        corners = P.transform(0.5 * np.asarray(dims) * np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)]))
        for q in (pts, corners):
            d = np.linalg.norm(q[:, None, :] - s[None, :, :3], axis=-1) - s[None, :, 3]
            assert d.min(axis=1).max() <= 1e-6
    m_in = fit_spheres_to_obstacle(Cuboid("b", [0, 0, 0, 1, 0, 0, 0], dims=[0.3, 0.1, 0.02]), 12)
    assert (np.abs(m_in[:, :3]) + m_in[:, 3:4] <= 0.5 * np.array([0.3, 0.1, 0.02]) + 1e-6).all()
