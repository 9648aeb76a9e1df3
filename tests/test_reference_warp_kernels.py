"""Scene collision against the REFERENCE's own Warp kernels.

``tests/golden/scene_warp_golden.npz`` holds inputs and outputs of the reference's unmodified
``sphere_obstacle_collision_kernel`` / ``swept_sphere_obstacle_collision_kernel`` / ``apply_speed_metric`` (and the
cuboid / fp16-ESDF accessors they call), executed thread by thread on the CPU through the Warp stand-in of
``tests/golden/warp_emulator`` (generator: ``tests/golden/make_scene_warp_golden.py``).  Cuboids (rotated, disabled,
two environments), voxel grids (rotated, disabled, spheres outside the grid), static and swept, speed metric,
negative-radius spheres, a stationary sphere, a sphere resting inside a box.

CPU: the C oracle reproduces them (distance to the last bit on these inputs, asserted at 1e-6); GPU: the HIP scene
kernel through the C ABI, at the path's 1e-5.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_warp_golden.npz")


def _cases():
    g = np.load(GOLD)
    for name, prm in zip([str(x) for x in g["case_names"]], g["case_params"]):
        w, eta, multi, swept, dt, has_c, has_v = prm
        scene = {}
        if has_c:
            scene.update({k: g[k] for k in g.files if k.startswith("cuboid_")})
        if has_v:
            scene.update({k: g[k] for k in g.files if k.startswith("voxel_")})
            scene["voxel_max_distance"] = float(g["voxel_max_distance"])
        yield (name, scene, float(w), float(eta), bool(multi), bool(swept), (float(dt) if dt > 0 else None), g["spheres"],
               g["env_query_idx"], g[f"{name}/distance"], g[f"{name}/gradient"])


CASES = list(_cases())
IDS = [c[0] for c in CASES]


def _compare(name, dist, grad, want_d, want_g, tol):
    assert dist.shape == want_d.shape
    scale_d, scale_g = max(1.0, float(want_d.max())), max(1.0, float(np.abs(want_g).max()))
    assert np.array_equal(dist > 0, want_d > 0), (name, "different spheres in collision")
    np.testing.assert_allclose(dist, want_d, rtol=0, atol=tol * scale_d, err_msg=name)
    np.testing.assert_allclose(grad[..., :3], want_g[..., :3], rtol=0, atol=tol * scale_g, err_msg=name)


def test_golden_covers_the_branches():
    """every case has hits and free spheres; the special spheres did what they were placed for"""
    for name, _scene, _w, _eta, _multi, swept, _dt, sp, _env, d, g in CASES:
        assert 10 < (d > 0).sum() < d.size - 10, name
        assert np.all(d[sp[..., 3] < 0] == 0) and np.all(g[sp[..., 3] < 0] == 0), name  # disabled spheres
        assert np.all(d[2, :, 2] == 0), name  # far outside every obstacle
        assert np.all(g[..., 3] == 0), name   # the fourth gradient slot is never written
    d = dict((c[0], c[9]) for c in CASES)
    assert np.all(d["cuboid_static"][1, :, 1] > 0) and np.ptp(d["cuboid_static"][1, :, 1]) == 0  # resting inside the table
    assert not np.array_equal(d["cuboid_swept"], d["cuboid_static"])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_the_reference_warp_kernels(case, oracle):
    name, scene, w, eta, multi, swept, dt, sp, env, want_d, want_g = case
    r = oracle.scene_collision(sp, scene, w, eta, env, multi, sweep=swept, enable_speed_metric=dt is not None,
                               speed_dt=dt if dt is not None else 0.02)
    _compare(name, r["distance"], r["gradient"], want_d, want_g, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_reproduces_the_reference_warp_kernels(case, device):
    import torch

    from curobo_amd.backends import collision as Cn
    from curobo_amd.scene import SceneData

    name, scene, w, eta, multi, swept, dt, sp, env, want_d, want_g = case
    sd = SceneData.from_arrays(scene, device)
    b, h, S, _ = sp.shape
    dist = torch.full((b, h, S), 7.0, device=device)  # the kernel must overwrite every entry
    grad = torch.full((b, h, S, 4), 7.0, device=device)
    Cn.sphere_obstacle_collision(
        dist, grad, torch.as_tensor(sp, device=device), sd.struct, torch.tensor([w], device=device),
        torch.tensor([eta], device=device), torch.as_tensor(env, device=device), b, h, S, multi, 3 if swept else 0,
        dt is not None, torch.tensor([dt if dt is not None else 0.02], device=device))
    torch.cuda.synchronize()
    _compare(name, dist.cpu().numpy(), grad.cpu().numpy(), want_d, want_g, 1e-5)
