"""Triangle-mesh obstacles on the device (csrc/mesh_bvh.hip: linear BVH build, closest point + sign, sphere-vs-mesh collision,
ESDF bake through the BVH) against the oracle's brute force over every triangle (oracle/curobo_oracle.c orc_mesh_sdf_raw).

The reference walks meshes with NVIDIA Warp's ``wp.mesh_query_point`` (data_mesh.py:630-700), which is neither in /root/reference
nor buildable for ROCm: the BVH walk itself is unpinned against Warp.  What is pinned: the same contract on the same inputs --
exact closest point (a BVH changes the order triangles are visited in, not the minimum), the sign of closed meshes, the
``max_distance`` rule -- and the reference's own regression case (tests/_src/collision/test_mesh_collision_sdf.py)."""

import numpy as np
import pytest
import torch

from conftest import load_model, sample_q
from test_oracle_mesh import box_shape, ell_shape, mesh_world, small_cube_case, sphere_shape, torus_shape

pytestmark = pytest.mark.gpu


SHAPES = {
    "box": lambda: box_shape([0.3, 0.5, 0.2], 3),            # 768 triangles, many coplanar: ties between triangles
    "sphere": lambda: sphere_shape(0.2, 32, 64),            # 3968
    "torus": lambda: torus_shape(0.22, 0.06, 64, 32),       # 4096, genus 1
    "ell": lambda: ell_shape(3),                            # non-convex polyhedron
    "tiny": lambda: box_shape([0.05, 0.05, 0.05], 0),       # 12 triangles: fewer than one wavefront, 4 leaves
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("leaf_size", [1, 4])
def test_bvh_query_is_the_brute_force_query(shape, leaf_size, oracle, device):
    from curobo_amd.backends.mesh import build_mesh_bvh, mesh_query

    v, f = SHAPES[shape]()
    mesh = build_mesh_bvh(v, f, device, leaf_size=leaf_size)
    assert mesh.n_tri == len(f)
    # the build keeps every triangle exactly once (sorted order): same multiset of (a, b - a, c - a)
    tri = mesh.tri.cpu().numpy().reshape(-1, 3, 4)[:, :, :3]
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    want = np.stack([a, b - a, c - a], 1).reshape(len(f), 9)
    got = tri.reshape(len(f), 9)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    # every node box holds its triangles; the root box is the mesh's bounding box
    box = mesh.node_box.cpu().numpy()
    np.testing.assert_array_equal(box[1, :3], v[f.reshape(-1)].min(0))
    np.testing.assert_array_equal(box[1, 4:7], v[f.reshape(-1)].max(0))
    rng = np.random.default_rng(5)
    ext = np.abs(v).max(0) + 0.15
    p = rng.uniform(-ext, ext, size=(6000, 3)).astype(np.float32)
    p[:200] = (v[f[rng.integers(0, len(f), 200), 0]] + rng.normal(size=(200, 3)) * 1e-3).astype(np.float32)  # hugging the surface
    for max_distance in (10.0, 0.08):
        ref_sdf, ref_grad = oracle.mesh_query(p, v, f, max_distance)
        sdf, grad = mesh_query(mesh, torch.as_tensor(p, device=device), max_distance)
        torch.cuda.synchronize()
        sdf, grad = sdf.cpu().numpy(), grad.cpu().numpy()
        # |distance|: same minimum over the same triangles (float rounding of the per-triangle distance only)
        np.testing.assert_allclose(np.abs(sdf), np.abs(ref_sdf), atol=2e-6, rtol=1e-5)
        # sign: ray-crossing parity on the device, winding number in the oracle: agree off the surface
        off = np.abs(ref_sdf) > 1e-5
        assert np.array_equal((sdf < 0)[off], (ref_sdf < 0)[off])
        assert (ref_sdf < 0).sum() > 20 or shape == "tiny"
        # cut-off: same set of points reports "nothing within max_distance"
        edge = np.abs(np.abs(ref_sdf) - max_distance) < 1e-5
        assert np.array_equal((sdf == np.float32(max_distance))[~edge], (ref_sdf == np.float32(max_distance))[~edge])
        # gradient: same closest point, except where two triangles are equally close up to rounding (medial axis)
        found = (ref_sdf != np.float32(max_distance)) & ~edge & (np.abs(ref_sdf) > 1e-4)
        err = np.abs(grad - ref_grad).max(-1)[found]
        assert (err > 1e-3).mean() < 0.01, (err > 1e-3).mean()
        assert np.median(err) < 1e-5


def test_reference_regression_small_mesh_cost_matches_cuboid(device):
    """the reference's own mesh regression test on the device: a 5 cm cube as mesh and as cuboid give the same costs -- first
    probe in collision, the other three exactly zero (two of them beyond half the bounding-box diagonal: the
    ``max(max_distance, query_distance)`` rule of data_mesh.py:668)"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, cuboid_scene_arrays

    v, f, sph = small_cube_case()
    pose = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    out = []
    for scene in (SceneData.from_arrays(None, device, meshes=[[{"name": "box", "vertices": v, "faces": f, "pose": pose}]]),
                  SceneData.from_arrays(cuboid_scene_arrays([[{"dims": [0.05] * 3, "pose": pose}]]), device)):
        dist, grad = torch.full((1, 1, 4), 7.0, device=device), torch.full((1, 1, 4, 4), 7.0, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
                                     torch.tensor([0.01], device=device), None, 1, 1, 4, False)
        torch.cuda.synchronize()
        out.append(dist.cpu().numpy().reshape(-1))
    assert np.allclose(out[0], out[1])
    assert out[0][0] > 0.0 and (out[0][1:] == 0.0).all()


def _trajectory_spheres(oracle, b, h, scale=0.7):
    model = load_model("franka")
    q0, q1 = sample_q(model, b, seed=11)[:, None], sample_q(model, b, seed=12)[:, None]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    sph = oracle.kinematics_forward((q0 * (1 - tt) + q1 * tt).reshape(b * h, -1) * scale, model.as_dict(), horizon=h)["robot_spheres"]
    return sph.reshape(b, h, -1, 4)


@pytest.mark.parametrize("sweep,speed", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("with_cuboids", [False, True])
def test_sphere_mesh_collision_matches_the_oracle(sweep, speed, with_cuboids, oracle, device):
    """robot spheres along trajectories against a mesh world (table, ball, torus, L prism; one disabled slot that shares a
    BVH): cost and gradient per sphere vs the oracle's scene restatement with the mesh kind; alone and on top of cuboids"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData, cuboid_scene_arrays

    world = mesh_world()
    sph = _trajectory_spheres(oracle, 24, 9)
    b, h, S, _ = sph.shape
    arrays = cuboid_scene_arrays([[{"dims": [0.1, 0.1, 1.5], "pose": [0.45, -0.3, 0.3, 1, 0, 0, 0]},
                                   {"type": "capsule", "radius": 0.07, "base": [0, 0, 0.0], "tip": [0, 0, 0.5],
                                    "pose": [-0.3, 0.5, 0.2, 0.9238795, 0.3826834, 0, 0]}]]) if with_cuboids else {}
    ref = oracle.scene_collision(sph, {**arrays, **mesh_scene_arrays(world)}, 3.0, 0.02, sweep=sweep, enable_speed_metric=speed, speed_dt=0.05)
    scene = SceneData.from_arrays(arrays or None, device, meshes=world)
    assert len(scene.meshes.meshes) == 4 and scene.meshes.max_n == 5
    dist, grad = torch.full((b, h, S), 5.0, device=device), torch.full((b, h, S, 4), 5.0, device=device)
    Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([3.0], device=device),
                                 torch.tensor([0.02], device=device), None, b, h, S, False, 3 if sweep else 0, speed,
                                 torch.tensor([0.05], device=device))
    torch.cuda.synchronize()
    d, g = dist.cpu().numpy(), grad.cpu().numpy()
    assert 0.03 < (ref["distance"] > 0).mean() < 0.9
    # hit set: exact except where a sphere grazes a surface within rounding
    graze = np.abs(d - ref["distance"]) < 2e-5
    assert np.array_equal((d > 0)[~graze], (ref["distance"] > 0)[~graze])
    scale = 20.0 if speed else 1.0  # the speed metric multiplies cost and gradient by the sphere speed (up to ~10 m/s here)
    np.testing.assert_allclose(d, ref["distance"], atol=2e-5 * scale, rtol=1e-4)
    bad = np.abs(g - ref["gradient"]).max(-1) > (2e-4 * scale + 1e-3 * np.abs(ref["gradient"]).max(-1))
    assert bad.mean() < 2e-3, bad.mean()  # (closest-point ties on coplanar triangles / the medial axis)


@pytest.mark.parametrize("sweep", [False, True])
@pytest.mark.parametrize("accumulate", [False, True])
def test_queued_mesh_launch_equals_the_one_kernel_launch(sweep, accumulate, oracle, device):
    """the launch-wide queue of the mesh launch (select + walk kernels: spheres inside a live mesh's bounding box queued from the
    head, the others from the tail; eight lanes per sphere) against the one-kernel form (a tree walk per lane) on the same
    buffers: the same spheres are live, costs agree to rounding, gradients except on closest-point ties"""
    from curobo_amd.backends import mesh as M
    from curobo_amd.scene import SceneData

    world = mesh_world()
    sph = torch.as_tensor(_trajectory_spheres(oracle, 24, 9), device=device)
    b, h, S, _ = sph.shape
    scene = SceneData.from_arrays(None, device, meshes=world)
    w, eta, dt = torch.tensor([3.0], device=device), torch.tensor([0.02], device=device), torch.tensor([0.05], device=device)
    outs = []
    for workspace in (None, False):
        dist, grad = torch.full((b, h, S), 0.25, device=device), torch.full((b, h, S, 4), 0.5, device=device)
        M.sphere_mesh_collision(dist, grad, sph, scene.struct.mesh_set, w, eta, None, b, h, S, False, 3 if sweep else 0, sweep, dt,
                                accumulate=accumulate, workspace=workspace)
        torch.cuda.synchronize()
        outs.append((dist.cpu().numpy(), grad.cpu().numpy()))
    (dq, gq), (d1, g1) = outs
    base = 0.25 if accumulate else 0.0
    live = d1 != base
    assert 0.03 < live.mean() < 0.9
    assert np.array_equal(dq != base, live)
    np.testing.assert_allclose(dq, d1, rtol=2e-5, atol=2e-6)
    bad = np.abs(gq - g1).max(-1) > 1e-4 + 1e-3 * np.abs(g1).max(-1)
    assert bad.mean() < 2e-3, bad.mean()


@pytest.mark.parametrize("h", [1, 2])
def test_many_short_trajectories_per_select_workgroup_with_environments(h, oracle, device):
    """few spheres per trajectory and an environment index per trajectory: a workgroup of the select kernel then meets more
    (trajectory, slot) pairs than its LDS table of obstacle slots holds and reads the slots per sphere instead -- same results"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    w0 = mesh_world()[0]
    envs = [[w0[0], w0[1]], [dict(w0[2]), dict(w0[3]), dict(w0[1], name="ball_b", mesh_name="ball")], [dict(w0[3]), dict(w0[0])]]
    sph = np.ascontiguousarray(_trajectory_spheres(oracle, 700, h)[:, :, 20:28])
    b, _, S, _ = sph.shape
    idx = (np.arange(b) % 3).astype(np.int32)
    scene = SceneData.from_arrays(None, device, meshes=envs)
    dist, grad = torch.full((b, h, S), 0.5, device=device), torch.full((b, h, S, 4), 0.5, device=device)
    Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([2.0], device=device),
                                 torch.tensor([0.02], device=device), torch.as_tensor(idx, device=device), b, h, S, True, 3 if h > 1 else 0, False, None)
    torch.cuda.synchronize()
    ref = oracle.scene_collision(sph, mesh_scene_arrays(envs), 2.0, 0.02, env_query_idx=idx, use_multi_env=True, sweep=h > 1)
    assert all((ref["distance"][k::3] > 0).any() for k in range(3))
    np.testing.assert_allclose(dist.cpu().numpy(), ref["distance"], atol=2e-5, rtol=1e-4)
    bad = np.abs(grad.cpu().numpy() - ref["gradient"]).max(-1) > 1e-4 + 1e-3 * np.abs(ref["gradient"]).max(-1)
    assert bad.mean() < 2e-3, bad.mean()


def test_mesh_slots_per_environment_pose_updates_and_enable(oracle, device):
    """two environments with different mesh sets, ``env_query_idx`` per trajectory; then move a mesh and switch one off:
    the BVH stays, the store's pose / enable rows change (reference MeshData.update_pose / enable_obstacle)"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    w0 = mesh_world()[0]
    envs = [[w0[0], w0[1]], [dict(w0[2]), dict(w0[3]), dict(w0[1], name="ball_b", mesh_name="ball")]]
    sph = _trajectory_spheres(oracle, 16, 7)
    b, h, S, _ = sph.shape
    idx = (np.arange(b) % 2).astype(np.int32)
    scene = SceneData.from_arrays(None, device, meshes=envs)
    assert len(scene.meshes.meshes) == 4  # "ball" is built once and used in both environments

    def run():
        dist, grad = torch.zeros(b, h, S, device=device), torch.zeros(b, h, S, 4, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([2.0], device=device),
                                     torch.tensor([0.02], device=device), torch.as_tensor(idx, device=device), b, h, S, True, 3, False, None)
        torch.cuda.synchronize()
        return dist.cpu().numpy()

    ref = oracle.scene_collision(sph, mesh_scene_arrays(envs), 2.0, 0.02, env_query_idx=idx, use_multi_env=True, sweep=True)["distance"]
    np.testing.assert_allclose(run(), ref, atol=2e-5, rtol=1e-4)
    assert (ref[0::2] > 0).any() and (ref[1::2] > 0).any()
    envs[1][0]["pose"] = [0.3, 0.0, 0.5, 1, 0, 0, 0]
    envs[1][1]["enable"] = False
    scene.meshes.update_pose("ring", envs[1][0]["pose"], env_idx=1)
    scene.meshes.set_enabled("ell", False, env_idx=1)
    ref2 = oracle.scene_collision(sph, mesh_scene_arrays(envs), 2.0, 0.02, env_query_idx=idx, use_multi_env=True, sweep=True)["distance"]
    assert np.abs(ref2 - ref).max() > 1e-3
    np.testing.assert_allclose(run(), ref2, atol=2e-5, rtol=1e-4)
    with pytest.raises(ValueError, match="not found"):
        scene.meshes.update_pose("ring", envs[1][0]["pose"], env_idx=0)


def test_consistent_gradient_mode_points_out_of_the_mesh_on_both_sides(oracle, device):
    """``MeshStore(gradient_mode=CONSISTENT_GRADIENT)``: the cost gradient of a box mesh equals the analytic cuboid's for centres
    outside the box as well (the reference's mesh query returns the opposite vector there, see scene/mesh.py); costs are the
    same in both modes"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import MeshStore, SceneData, cuboid_scene_arrays

    dims, pose = [0.3, 0.4, 0.5], [0.3, 0.1, 0.4, np.cos(0.4), 0, 0, np.sin(0.4)]
    v, f = box_shape(dims)
    rng = np.random.default_rng(7)
    n = 4096
    sph = np.concatenate([rng.uniform([-0.1, -0.3, 0.0], [0.7, 0.5, 0.8], size=(n, 3)), rng.uniform(0.02, 0.08, size=(n, 1))], -1)
    sph = sph.astype(np.float32).reshape(1, 1, n, 4)
    out = {}
    for key, scene in (
            ("cuboid", SceneData.from_arrays(cuboid_scene_arrays([[{"dims": dims, "pose": pose}]]), device)),
            ("reference", SceneData.from_arrays(None, device, meshes=[[{"name": "b", "vertices": v, "faces": f, "pose": pose}]])),
            ("consistent", SceneData.from_arrays(None, device, meshes=MeshStore([[{"name": "b", "vertices": v, "faces": f, "pose": pose}]], device,
                                                                                gradient_mode=MeshStore.CONSISTENT_GRADIENT)))):
        dist, grad = torch.zeros(1, 1, n, device=device), torch.zeros(1, 1, n, 4, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([1.0], device=device),
                                     torch.tensor([0.02], device=device), None, 1, 1, n, False)
        torch.cuda.synchronize()
        out[key] = (dist.cpu().numpy().reshape(-1), grad.cpu().numpy().reshape(n, 4)[:, :3])
    hit = out["cuboid"][0] > 1e-4
    assert hit.sum() > 300
    np.testing.assert_allclose(out["reference"][0], out["cuboid"][0], atol=2e-6, rtol=1e-4)
    assert np.array_equal(out["reference"][0], out["consistent"][0])
    # off the medial axis (where the closest face is unique) the consistent mode is the cuboid's gradient
    agree = np.abs(out["consistent"][1] - out["cuboid"][1]).max(-1) < 1e-3
    assert agree[hit].mean() > 0.9
    # the reference's vector: the cuboid's for centres inside the box, its opposite for centres outside
    flipped = np.abs(out["reference"][1] + out["cuboid"][1]).max(-1) < 1e-3
    same = np.abs(out["reference"][1] - out["cuboid"][1]).max(-1) < 1e-3
    sel = hit & agree & (np.abs(out["cuboid"][1]).max(-1) > 1e-2)
    assert (flipped & sel).sum() > 100 and (same & sel).sum() > 100
    assert ((flipped ^ same) | ~sel).all()


def test_esdf_bake_through_the_bvh_equals_the_all_triangles_bake(device):
    """mesh -> fp16 ESDF grid: the BVH bake and the bake that visits every triangle per voxel (csrc/mesh_bake.hip, pinned against
    the NumPy mesh signed distance in test_gpu_kernels.py) fill the same grid"""
    from curobo_amd.backends.collision import mesh_esdf_bake
    from curobo_amd.backends.mesh import build_mesh_bvh, mesh_esdf_bake_bvh

    v, f = torus_shape(0.22, 0.06, 64, 32)
    nx, ny, nz, vs = 48, 48, 24, 0.0125
    xf = [1, 0, 0, 0.01, 0, 1, 0, -0.02, 0, 0, 1, 0.005]
    a = torch.zeros(nx * ny * nz, dtype=torch.float16, device=device)
    b = torch.zeros_like(a)
    mesh_esdf_bake(a, torch.as_tensor(v, device=device), torch.as_tensor(f, device=device), (nx, ny, nz), vs, xf, max_distance=0.1)
    mesh_esdf_bake_bvh(b, build_mesh_bvh(v, f, device), (nx, ny, nz), vs, xf, max_distance=0.1)
    torch.cuda.synchronize()
    a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
    assert (a < 0).sum() > 500 and (a == np.float32(np.float16(0.1))).sum() > 500
    assert np.abs(a - b).max() <= 2.0 ** -11 * 0.1 * 2  # one fp16 ulp at the largest magnitude
    assert (a != b).mean() < 0.01


def test_rollout_with_a_mesh_scene_uses_the_kernel_sequence(oracle, device):
    """a mesh world through the collision rollout (the fused kernel does not walk meshes: ``fused_available`` is off and the
    launch sequence with the mesh pass runs): the scene cost of the rollout's own spheres equals the oracle's, the gradient
    reaches the knots"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(arrays, device, meshes=mesh_world())
    B = 16
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_self_collision=False))
    assert not ro.fused_available()
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    x = torch.as_tensor(seed_knots(model, B, 12, seed=11), device=device).reshape(B, -1)
    cost, grad = [t.clone() for t in ro.cost_and_gradient(x)]
    torch.cuda.synchronize()
    sph = ro.robot_spheres.cpu().numpy()
    cfg = ro.cfg
    ref = oracle.scene_collision(sph, {**arrays, **mesh_scene_arrays(mesh_world())}, cfg.scene_collision_weight, cfg.activation_distance,
                                 sweep=True, enable_speed_metric=True, speed_dt=cfg.traj_dt)
    only_cub = oracle.scene_collision(sph, arrays, cfg.scene_collision_weight, cfg.activation_distance, sweep=True, enable_speed_metric=True,
                                      speed_dt=cfg.traj_dt)
    want = ref["distance"].reshape(B, -1).astype(np.float64).sum(-1)
    assert (want > only_cub["distance"].reshape(B, -1).sum(-1) * 1.01 + 1.0).any(), "the meshes must matter"
    # per sphere: equal up to rounding, except the few whose sweep takes a different branch (the swept cost jumps where a
    # sample's penetration crosses zero: test_gpu_rollout.py discusses it); the trajectory sums follow
    d = ro.scene_dist.cpu().numpy()
    bad = np.abs(d - ref["distance"]) > 2e-5 * cfg.scene_collision_weight * 20 + 1e-3 * np.abs(ref["distance"])
    assert bad.mean() < 1e-3, bad.mean()
    np.testing.assert_allclose(cost.cpu().numpy(), want, rtol=2e-2)
    assert (np.abs(cost.cpu().numpy() - want) < 2e-3 * want + 1e-2).mean() > 0.8
    assert float(grad.abs().max()) > 0.0


@pytest.mark.parametrize("with_cuboids", [True, False], ids=["cuboids+meshes", "meshes only"])
def test_trajopt_rollout_in_a_mesh_world_against_the_oracle_composition(with_cuboids, oracle, device):
    """The full trajopt rollout (tool pose + c-space + self collision + scene) in a world with meshes.  ``fused_available`` is
    off (the fused launch does not walk meshes) and the kernel sequence runs with the mesh launch adding its share to the
    cuboid kernel's: cost and d cost / d knots against the oracle's composition of the stages with the mesh kind in its scene
    restatement, and replayed from a hipGraph.
    (Measured and not kept, twice: the fused launch for every other term + the meshes' share by its own chain on a side stream --
    B-spline samples -> FK -> mesh launch -> FK VJP -> B-spline VJP, added to the fused result; equal to this sequence to 3e-7.
    Round 4: SLOWER at 1024 rollouts, 575 us against 540 us in the bench's mesh world.  Round 6, at a planner's sizes: the same --
    16 / 32 / 64 / 128 rollouts 116 / 135 / 143 / 554 us against the sequence's 115 / 137 / 149 / 569 us
    (tools/r06/mesh_rollout_forms.py as it then was): FK -> mesh launch -> FK VJP is the critical path of both.)"""
    from oracle.oracle import mesh_scene_arrays
    from oracle_compose import trajopt_cost_and_gradient

    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    arrays = cuboid_scene_arrays(c2_world()) if with_cuboids else {}
    B = 32
    x = torch.as_tensor(seed_knots(model, B, 12, seed=11), device=device).reshape(B, -1)
    start = torch.as_tensor(start_configuration(model), device=device)
    fused_ro = ro = TrajOptRollout(kin, SceneData.from_arrays(arrays or None, device, meshes=mesh_world()), B, TrajOptRolloutCfg())
    assert not ro.fused_available()
    ro.update_start_state(start)
    ro.cost_and_gradient(x)
    cf, gf = [t.clone() for t in ro.cost_and_gradient(x)]
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cg, gg = ro.cost_and_gradient(x)
    cg.zero_(), gg.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, cf) and torch.equal(gg, gf), "hipGraph replay == eager"
    only = TrajOptRollout(kin, SceneData.from_arrays(arrays, device) if with_cuboids else None, B, TrajOptRolloutCfg())
    only.update_start_state(start)
    assert bool((cf > only.cost_and_gradient(x)[0] * 1.01 + 1.0).any()), "the meshes must matter"
    # The oracle's composition of the stages.  Its scene stage is evaluated on the spheres the device computed: the swept cost
    # of a sphere that does not move counts its centre once, twice or three times depending on whether its neighbours are
    # EXACTLY where it is (wp_sweep_collision_kernel.py:197-203), and the link next to the base sits still over the first
    # points -- the last bit of its FK decides (test_gpu_parity_benchmarked.py holds that rule per sphere).
    cfg = ro.cfg
    sph = fused_ro.robot_spheres.cpu().numpy()
    ref = trajopt_cost_and_gradient(oracle, model, cfg, x.cpu().numpy().reshape(B, 12, -1), start_configuration(model),
                                    scene_arrays={**arrays, **mesh_scene_arrays(mesh_world())}, scene_spheres=sph)
    kw = dict(sweep=cfg.use_sweep, enable_speed_metric=cfg.use_sweep and cfg.use_speed_metric, speed_dt=cfg.traj_dt)
    mesh_ref = oracle.scene_collision(sph, mesh_scene_arrays(mesh_world()), cfg.scene_collision_weight, cfg.scene_activation_distance, **kw)
    if not with_cuboids:  # (scene_dist holds the meshes' share alone)
        d = fused_ro.scene_dist.cpu().numpy()
        bad = np.abs(d - mesh_ref["distance"]) > 2e-5 * cfg.scene_collision_weight * 20 + 1e-3 * np.abs(mesh_ref["distance"])
        assert bad.mean() < 1e-3, bad.mean()
    want, got = ref["cost"], cf.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-2)
    # trajectories in which no sweep sample took another branch (a sample's penetration crossing zero within rounding):
    # cost to 1e-5, gradient to 5e-4 of its scale
    close = np.abs(got - want) < 1e-5 * np.abs(want)
    print(f"[mesh world] vs the oracle: {int(close.sum())} of {B} trajectories to 1e-5, worst {float((np.abs(got - want) / np.abs(want)).max()):.2e}")
    assert close.mean() >= 0.75
    gk = ref["grad_knots"].reshape(B, -1)
    np.testing.assert_allclose(gf.cpu().numpy()[close], gk[close], rtol=5e-4, atol=5e-4 * np.abs(gk[close]).max())


def test_collision_checker_from_a_scene_config_with_a_mesh_file(tmp_path, oracle, device):
    """``scene_model={"cuboid": ..., "mesh": {name: {"file_path": *.obj, "pose", "scale"}}}`` (the reference's SceneCfg format)
    through ``RobotCollisionChecker``: the OBJ is read, scaled, placed; robot-vs-scene distances equal the oracle's"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.collision_checking import RobotCollisionChecker, RobotCollisionCheckerCfg
    from curobo_amd.scene import cuboid_scene_arrays

    v, f = torus_shape(0.11, 0.03)
    path = tmp_path / "ring.obj"
    with open(path, "w") as fh:
        fh.write("# ring\n" + "".join(f"v {a:.7f} {b:.7f} {c:.7f}\n" for a, b, c in v) + "".join(f"f {a + 1} {b + 1}/1 {c + 1}//2\n" for a, b, c in f))
    pose = [0.4, 0.0, 0.45, 0.9238795, 0.3826834, 0, 0]
    cfg = {"cuboid": {"table": {"dims": [2.2, 2.2, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}},
           "mesh": {"ring": {"file_path": str(path), "pose": pose, "scale": [2.0, 2.0, 2.0]}}}
    chk = RobotCollisionChecker(RobotCollisionCheckerCfg.load_from_config("franka.yml", cfg, 0.02, device=device))
    assert chk.scene.meshes is not None and chk.scene.meshes.meshes[0].n_tri == len(f)
    model = load_model("franka")
    q = sample_q(model, 256, seed=4, scale=0.7)
    sph = oracle.kinematics_forward(q, model.as_dict(), horizon=1)["robot_spheres"].reshape(256, 1, -1, 4)
    v2 = np.loadtxt([ln[2:] for ln in open(path) if ln.startswith("v ")], dtype=np.float32) * 2.0
    arrays = {**cuboid_scene_arrays([[{"dims": [2.2, 2.2, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}]]),
              **mesh_scene_arrays([[{"name": "ring", "vertices": v2, "faces": f, "pose": pose}]])}
    ref = oracle.scene_collision(sph, arrays, 1.0, 0.02)["distance"].reshape(256, -1)
    only_table = oracle.scene_collision(sph, cuboid_scene_arrays([[{"dims": [2.2, 2.2, 0.2], "pose": [0, 0, -0.1, 1, 0, 0, 0]}]]), 1.0, 0.02)
    assert (ref.sum(-1) > only_table["distance"].reshape(256, -1).sum(-1) + 1e-3).sum() > 5
    d_scene, _ = chk.get_scene_self_collision_distance_from_joints(torch.as_tensor(q, device=device))
    torch.cuda.synchronize()
    np.testing.assert_allclose(d_scene.detach().cpu().numpy().reshape(256, -1), ref, atol=2e-5, rtol=1e-4)


def test_seed_shards_over_a_mesh_scene_own_their_launch_workspaces(oracle, device):
    """PipelinedLBFGS with two seed shards (two streams, the parallel branches of one captured graph) over rollouts of EQUAL
    size in a mesh scene.  The queued mesh launch keeps a counter and a sphere queue per launch; cached by size alone they
    were shared by the shards -- one shard's select kernel cleared or filled the queue the other's walk kernel was reading
    (ADVICE round 4).  The workspace is now kept per output buffer: the shards must take the iterates of the one batch."""
    from curobo_amd.backends import mesh as mesh_backend
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg, PipelinedLBFGS
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device, meshes=mesh_world())
    cfg = CollisionRolloutCfg()
    seeds = 16
    ocfg = LBFGSOptCfg(num_problems=seeds, inner_iters=4, num_iters=8)
    nls = len(ocfg.line_search_scale)
    start = torch.as_tensor(start_configuration(model), device=device)
    bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
    made = []

    def make(batch):
        ro = CollisionRollout(kin, scene, batch, cfg)
        assert not ro.fused_available()  # a mesh world runs the kernel sequence with the queued mesh launch
        ro.update_start_state(start)
        made.append(ro)
        return ro.cost_and_gradient

    x0 = torch.as_tensor(seed_knots(model, seeds, cfg.n_knots, seed=4, spread=0.4), device=device)
    one = LBFGSOpt(ocfg, make(seeds * nls), cfg.n_knots, kin.num_dof, bounds, device)
    ref = one.optimize(x0).clone()
    ref_cost = one.best_cost.clone()
    n_ws = len(mesh_backend._WORKSPACES)
    pipe = PipelinedLBFGS(ocfg, make, cfg.n_knots, kin.num_dof, bounds, device, n_shards=2)
    for _ in range(3):  # (replays of the captured graph: the race needed both shards in flight)
        got = pipe.optimize(x0)
    torch.cuda.synchronize()
    assert len(made) == 3 and made[1].batch_size == made[2].batch_size, "two shards of equal size"
    assert len(mesh_backend._WORKSPACES) >= n_ws + 2, "each shard's output buffer has its own launch workspace"
    assert torch.isfinite(ref_cost).all() and float(ref_cost.max()) > 0
    torch.testing.assert_close(pipe.best_cost, ref_cost, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("horizon", [1, 8])
def test_captured_mesh_launch_replays_like_the_eager_launch(horizon, device):
    """the queued launch inside a hipGraph, replayed as a solver replays it (IKSolver's ranking graph: horizon 1).  The
    launch used to clear its queue counters with hipMemsetAsync; the captured 16-byte memset node replayed once and faulted
    at the SECOND replay (tools/r06/mesh_graph_replay.py) -- the counters are now cleared by a kernel.  Every replay on moved
    spheres gives the eager launch's numbers to the bit and leaves the launch's own counters."""
    from curobo_amd.backends import mesh as M
    from curobo_amd.scene import MeshStore

    store = MeshStore(mesh_world(), device)
    b, S = 32, 65
    g = torch.Generator().manual_seed(3)
    sph = torch.cat([torch.rand(b, horizon, S, 3, generator=g) * 1.6 - 0.8, torch.full((b, horizon, S, 1), 0.05)], -1).to(device)
    w, eta = torch.tensor([1.0], device=device), torch.tensor([0.01], device=device)
    out = [(torch.zeros(b, horizon, S, device=device), torch.zeros(b, horizon, S, 4, device=device)) for _ in range(2)]

    def launch(k):
        M.sphere_mesh_collision(out[k][0], out[k][1], sph, store.struct, w, eta, None, b, horizon, S, False, 0, False, None, accumulate=False)

    launch(0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch(0)
    ws = [next(iter(o[0]._curobo_mesh_ws.values())) if hasattr(o[0], "_curobo_mesh_ws") else None for o in out]
    for rep in range(5):
        sph[..., :3] += 0.02
        graph.replay()
        launch(1)
        torch.cuda.synchronize()
        assert float(out[1][0].sum()) > 0
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), rep
        ws[1] = next(iter(out[1][0]._curobo_mesh_ws.values()))
        assert torch.equal(ws[0][:16], ws[1][:16]), rep  # (the counters of this launch alone: cleared at its start)


@pytest.mark.parametrize("exit_early", [False, True])
def test_ik_solver_in_a_mesh_world(exit_early, oracle, device):
    """IKSolver over a scene of meshes (the kernel sequence with the queued mesh launch inside the optimiser's and the ranking
    graph), solved three times (graph replays): solutions reach their goals and are free of collision by the oracle's brute force
    over every triangle.  Consistent mesh gradient, as the scenes the solvers build have it (scene/config.py)."""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import MeshStore, SceneData
    from curobo_amd.solver import IKSolver, IKSolverCfg

    model = load_model("franka")
    md = model.as_dict()
    kin = KinematicsParams.from_model(model, device)
    world = mesh_world()
    arrays = mesh_scene_arrays(world)
    scene = SceneData.from_arrays(None, device, meshes=MeshStore(world, device, gradient_mode=MeshStore.CONSISTENT_GRADIENT))
    P = 12
    cand = sample_q(model, 600, seed=13, scale=0.8)
    fk = oracle.kinematics_forward(cand, md)
    sph = fk["robot_spheres"].reshape(600, 1, -1, 4)
    free = (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0) & \
        (oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0)
    assert 0.1 < free.mean() < 0.9, "the meshes must rule out a good share of the samples"
    sel = np.nonzero(free)[0][:P]
    gp, gq = fk["link_pos"][sel, 0], fk["link_quat"][sel, 0]
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=32, exit_early=exit_early))
    for rep in range(3):
        res = solver.solve_pose(torch.as_tensor(gp), torch.as_tensor(gq))
        torch.cuda.synchronize()
        succ = res.success.cpu().numpy()
        assert succ.mean() >= 0.8, f"IK success rate {succ.mean():.2f} (solve {rep})"
        qs = res.solution.cpu().numpy()[succ]
        chk = oracle.kinematics_forward(qs, md)
        np.testing.assert_allclose(chk["link_pos"][:, 0], gp[succ], atol=5e-3)
        s2 = chk["robot_spheres"].reshape(len(qs), 1, -1, 4)
        assert (oracle.self_collision(s2, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all()
        assert (oracle.scene_collision(s2, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0).all()
    if not exit_early:
        assert solver.optimizer_ran


# ------------------------------------------------------------------------------------------------ cell lists (round 6)
def _mesh_launch(device, world, sph, sweep, cells, **store_kw):
    from curobo_amd.backends import mesh as M
    from curobo_amd.scene import MeshStore

    store = MeshStore(world, device, cells=cells, **store_kw)
    b, h, S, _ = sph.shape
    w, eta, dt = torch.tensor([3.0], device=device), torch.tensor([0.02], device=device), torch.tensor([0.05], device=device)
    dist, grad = torch.full((b, h, S), 0.25, device=device), torch.full((b, h, S, 4), 0.5, device=device)
    M.sphere_mesh_collision(dist, grad, sph, store.struct, w, eta, None, b, h, S, False, 3 if sweep else 0, sweep, dt, accumulate=False)
    torch.cuda.synchronize()
    owner = dist._base if dist._base is not None else dist
    ws = next(iter(owner._curobo_mesh_ws.values()))
    counters = ws[:16].view(torch.int32).cpu().numpy()  # [heavy, handed to the tree walk, light, handed to a workgroup of its own]
    return dist.cpu().numpy(), grad.cpu().numpy(), counters, store


@pytest.mark.parametrize("sweep", [False, True])
@pytest.mark.parametrize("cell_size", [0.02, 0.05])
def test_cell_list_launch_is_the_tree_walk_launch(sweep, cell_size, oracle, device):
    """The queued mesh launch through the distance-sorted cell lists (select -> cell-list kernel -> tree walk of what the lists
    cannot answer) against the same launch over meshes built WITHOUT lists (select -> tree walk): the closest point is the
    same minimum over the same fp32 point-triangle distances, so costs are equal to the bit except where a sweep sample's
    branch turns on a tie; and on these closed meshes (next to) no sphere is handed to the tree walk."""
    world = mesh_world()
    sph = torch.as_tensor(_trajectory_spheres(oracle, 96, 9, scale=0.5), device=device)  # (steps of a few centimetres)
    d_c, g_c, cnt_c, store = _mesh_launch(device, world, sph, sweep, {"cell_size": cell_size})
    d_w, g_w, cnt_w, _ = _mesh_launch(device, world, sph, sweep, False)
    info = [m.cells_info for m in store.meshes]
    assert all(i is not None and i["entries"] > i["cells"] for i in info)
    live_c, live_w = int(cnt_c[0]) + int(cnt_c[2]), int(cnt_w[0]) + int(cnt_w[2])
    # (the select kernel drops spheres whose cell is wholly outside and farther than their reach: fewer live spheres)
    assert 100 < live_c <= live_w
    # (a sphere that moves more than the grid's pad per step can still reach past it: at most a couple here)
    assert cnt_w[1] == 0 and cnt_c[1] <= 2, (cnt_c, info)
    assert 0.002 < (d_w > 0).mean() < 0.9
    assert np.array_equal(d_c > 0, d_w > 0)
    assert (d_c != d_w).mean() < 1e-4
    np.testing.assert_allclose(d_c, d_w, rtol=2e-6, atol=1e-7)
    bad = np.abs(g_c - g_w).max(-1) > 1e-5 + 1e-4 * np.abs(g_w).max(-1)
    assert bad.mean() < 2e-3, bad.mean()   # (the medial axis: two triangles equally close)


def test_cell_lists_fall_back_to_the_walk_beyond_the_grid_and_without_lists(oracle, device):
    """a grid with no pad and a cap of one triangle per list: nearly every query is outside the grid or in a cell without a list, so
    nearly every live sphere goes through queue2 to the tree walk -- and the results are still those of the walk launch.  Fast
    trajectories (steps of decimetres: the sweep's reach exceeds any pad) do the same to the default grid."""
    world = mesh_world()
    sph = torch.as_tensor(_trajectory_spheres(oracle, 24, 9), device=device)
    d_w, g_w, cnt_w, _ = _mesh_launch(device, world, sph, True, False)
    for cells, least in (({"cell_size": 0.05, "pad": 0.0, "gather_cap": 1}, 0.5), (None, 0.0)):
        d_c, g_c, cnt_c, _ = _mesh_launch(device, world, sph, True, cells)
        live = int(cnt_c[0]) + int(cnt_c[2])
        # (outside the grid: to the tree walk, word 1; a cell without a list: to a workgroup of its own, word 3)
        assert cnt_c[1] + cnt_c[3] >= least * live and (least == 0.0 or (cnt_c[1] > 0 and cnt_c[3] > 0))
        np.testing.assert_allclose(d_c, d_w, rtol=2e-6, atol=1e-7)
        bad = np.abs(g_c - g_w).max(-1) > 1e-5 + 1e-4 * np.abs(g_w).max(-1)
        assert bad.mean() < 2e-3


@pytest.mark.parametrize("sweep", [False, True])
def test_spheres_in_the_middle_of_a_ball_get_a_workgroup_each(sweep, oracle, device):
    """Points about equally far from every triangle (the middle of a ball of 3 968 triangles, the axis of a torus' tube): the
    cells there have no list (more candidates than the build gathers) or a prefix that is the whole list, and no tree prunes --
    eight lanes held the launch for 430 us with FIVE such spheres (tools/r06/mesh_fallback_probe.py).  The cell-list kernel
    hands them to ``sphere_mesh_wide_kernel`` (a workgroup per sphere, every leaf within the a-priori bound tested by 256
    lanes): same distances as the tree walk launch, none of them sent to the walk, and as the oracle's brute force has them."""
    from oracle.oracle import mesh_scene_arrays

    vs, fs = sphere_shape(0.2, 32, 64)
    world = [[{"name": "ball", "vertices": vs, "faces": fs, "pose": [0.1, -0.2, 0.3, 0.9238795, 0, 0.3826834, 0]}]]
    rng = np.random.default_rng(5)
    b, h, S = 6, 5, 16
    c = np.array([0.1, -0.2, 0.3], np.float32)
    sph = np.zeros((b, h, S, 4), np.float32)
    sph[..., :3] = c + rng.normal(size=(b, 1, S, 3)).astype(np.float32) * np.array([0.004, 0.004, 0.06], np.float32)  # around the middle, and out along z
    sph[..., :3] += (np.arange(h, dtype=np.float32)[None, :, None, None] - 2) * rng.normal(size=(b, 1, S, 3)).astype(np.float32) * 0.01
    sph[..., 3] = 0.03
    t = torch.as_tensor(sph, device=device)
    d_c, g_c, cnt_c, store = _mesh_launch(device, world, t, sweep, None)
    d_w, g_w, cnt_w, _ = _mesh_launch(device, world, t, sweep, False)
    assert store.meshes[0].cells_info["max_list"] > 16 * 32 or store.meshes[0].cells_info["cells_without_list"] > 0
    assert cnt_c[3] > 10 and cnt_c[1] == 0 and cnt_w[1] == 0, cnt_c
    assert (d_w > 0).all()  # (every sphere is inside the ball)
    np.testing.assert_allclose(d_c, d_w, rtol=2e-6, atol=1e-7)
    assert (d_c != d_w).mean() < 0.02
    bad = np.abs(g_c - g_w).max(-1) > 1e-5 + 1e-4 * np.abs(g_w).max(-1)
    assert bad.mean() < 0.05, bad.mean()  # (near the middle the closest triangle is one of many within rounding of each other)
    ref = oracle.scene_collision(sph, mesh_scene_arrays(world), 3.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
    np.testing.assert_allclose(d_c, ref["distance"], rtol=1e-4, atol=2e-5 * (20.0 if sweep else 1.0))


def _open_fixtures():
    from test_oracle_mesh_sign import face_normals, plate, with_flipped, without_faces_facing

    vb, fb = box_shape([0.3, 0.5, 0.2], 2)
    n = face_normals(vb, fb)
    return {
        "open_box(+z missing)": without_faces_facing(vb, fb, [0, 0, 1]),
        "open_box(-z missing)": without_faces_facing(vb, fb, [0, 0, -1]),
        "single_sided_plate": plate(),
        "box_one_flipped_face": with_flipped(vb, fb, [np.flatnonzero(n[:, 0] > 0.999)[3]]),
    }


@pytest.mark.parametrize("name", ["open_box(+z missing)", "open_box(-z missing)", "single_sided_plate", "box_one_flipped_face"])
def test_open_and_flipped_meshes_take_the_reference_ray_sign(name, oracle, device):
    """Meshes that are not closed or not consistently oriented are signed on the device by the reference's rule -- Warp's three
    rays (+x, +y, +z; inside iff every ray's nearest hit is a back face), restated in the oracle in fp64 (``set_mesh_sign_rule("rays")``;
    tests/test_oracle_mesh_sign.py) -- through both query paths: the point query (tree walk per lane) and the sphere launch (cell
    lists).  On these fixtures the closest-feature rule would give another answer on up to half of the probes."""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.backends import collision as Cn
    from curobo_amd.backends.mesh import SIGN_WARP_RAYS, build_mesh_bvh, mesh_query
    from curobo_amd.scene import SceneData

    v, f = _open_fixtures()[name]
    mesh = build_mesh_bvh(v, f, device)
    assert mesh.sign_rule == SIGN_WARP_RAYS and mesh.struct.sign_rule == 1
    rng = np.random.default_rng(11)
    p = rng.uniform(v.min(0) - 0.1, v.max(0) + 0.1, size=(4000, 3)).astype(np.float32)
    oracle.set_mesh_sign_rule("rays")
    try:
        ref_sdf, _ = oracle.mesh_query(p, v, f, 10.0)
        sdf, _ = mesh_query(mesh, torch.as_tensor(p, device=device), 10.0)
        torch.cuda.synchronize()
        sdf = sdf.cpu().numpy()
        np.testing.assert_allclose(np.abs(sdf), np.abs(ref_sdf), atol=2e-6, rtol=1e-5)
        # a probe whose ray passes within rounding of an edge or whose nearest hit is at rounding distance may differ: fp32 rays
        # on the device against fp64 in the oracle -- a handful in thousands
        off = np.abs(ref_sdf) > 1e-4
        differ = ((sdf < 0) != (ref_sdf < 0)) & off
        assert differ.mean() < 2e-3, differ.mean()
        if name != "single_sided_plate":
            assert (ref_sdf < 0).sum() >= (0 if name == "open_box(+z missing)" else 50)
        # the sphere launch over the same mesh at a pose
        pose = [0.05, -0.02, 0.3, 0.9659258, 0, 0.2588190, 0]
        world = [[{"name": "m", "vertices": v, "faces": f, "pose": pose}]]
        sph = np.concatenate([rng.uniform([-0.3, -0.4, 0.0], [0.4, 0.4, 0.6], size=(6000, 3)), rng.uniform(0.01, 0.05, size=(6000, 1))], 1)
        sph = sph.astype(np.float32).reshape(8, 5, 150, 4)
        for sweep in (False, True):
            ref = oracle.scene_collision(sph, mesh_scene_arrays(world), 2.0, 0.02, sweep=sweep)
            scene = SceneData.from_arrays(None, device, meshes=world)
            assert scene.meshes.meshes[0].sign_rule == SIGN_WARP_RAYS
            dist, grad = torch.zeros(8, 5, 150, device=device), torch.zeros(8, 5, 150, 4, device=device)
            Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([2.0], device=device),
                                         torch.tensor([0.02], device=device), None, 8, 5, 150, False, 3 if sweep else 0, False, None)
            torch.cuda.synchronize()
            d = dist.cpu().numpy()
            bad = np.abs(d - ref["distance"]) > 2e-5 + 1e-4 * np.abs(ref["distance"])
            assert (ref["distance"] > 0).mean() > 0.02
            assert bad.mean() < (4e-3 if sweep else 2e-3), bad.mean()
    finally:
        oracle.set_mesh_sign_rule("winding")


def test_cell_lists_that_exceed_their_budget_coarsen_or_are_left_out(oracle, device):
    """``build_mesh_cells(max_entries=)``: lists that do not fit get coarser cells; a mesh whose lists never fit keeps the tree walk
    alone -- the launch's results are the same either way"""
    from curobo_amd.backends.mesh import build_mesh_bvh

    v, f = torus_shape(0.22, 0.06, 64, 32)
    full = build_mesh_bvh(v, f, device)
    assert full.cells_info["entries"] > 50_000
    small = build_mesh_bvh(v, f, device, cells={"max_entries": full.cells_info["entries"] // 2})
    assert small.cell_start is not None and small.cells_info["cell_size"] > full.cells_info["cell_size"]
    assert small.cells_info["entries"] <= full.cells_info["entries"] // 2
    none = build_mesh_bvh(v, f, device, cells={"max_entries": 10})
    assert none.cell_start is None and none.struct.cell_start is None and "skipped" in none.cells_info
    world = [[{"name": "ring", "vertices": v, "faces": f, "pose": [0.1, 0.45, 0.45, 0.9238795, 0.3826834, 0, 0]}]]
    sph = torch.as_tensor(_trajectory_spheres(oracle, 24, 9, scale=0.6), device=device)
    outs = [_mesh_launch(device, world, sph, True, c)[0] for c in ({}, {"max_entries": full.cells_info["entries"] // 2}, {"max_entries": 10})]
    assert (outs[0] > 0).sum() > 10
    np.testing.assert_allclose(outs[1], outs[0], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(outs[2], outs[0], rtol=2e-6, atol=1e-7)


def test_more_than_thirty_two_mesh_slots_run_as_slot_groups(oracle, device):
    """a sphere's live slots are a 32-bit mask: 40 mesh obstacles (one shared BVH, 40 poses) run as two slot groups into the same
    buffers, through the cell lists -- against the oracle's scene restatement"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    v, f = box_shape([0.08, 0.08, 0.08], 1)
    rng = np.random.default_rng(4)
    world = [[{"name": f"cube{i}", "mesh_name": "cube", "vertices": v, "faces": f,
               "pose": [float(x) for x in rng.uniform([-0.6, -0.6, 0.1], [0.6, 0.6, 0.9])] + [1, 0, 0, 0], "enable": i % 7 != 3}
              for i in range(40)]]
    sph = _trajectory_spheres(oracle, 16, 5, scale=0.5)
    b, h, S, _ = sph.shape
    scene = SceneData.from_arrays(None, device, meshes=world)
    assert scene.meshes.max_n == 40 and len(scene.meshes.meshes) == 1 and scene.meshes.meshes[0].cell_start is not None
    for sweep in (False, True):
        ref = oracle.scene_collision(sph, mesh_scene_arrays(world), 2.0, 0.02, sweep=sweep)
        dist, grad = torch.full((b, h, S), 3.0, device=device), torch.full((b, h, S, 4), 3.0, device=device)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=device), scene.struct, torch.tensor([2.0], device=device),
                                     torch.tensor([0.02], device=device), None, b, h, S, False, 3 if sweep else 0, False, None)
        torch.cuda.synchronize()
        assert (ref["distance"] > 0).mean() > 0.01
        np.testing.assert_allclose(dist.cpu().numpy(), ref["distance"], atol=2e-5, rtol=1e-4)
        bad = np.abs(grad.cpu().numpy() - ref["gradient"]).max(-1) > 1e-4 + 1e-3 * np.abs(ref["gradient"]).max(-1)
        assert bad.mean() < 2e-3, bad.mean()
