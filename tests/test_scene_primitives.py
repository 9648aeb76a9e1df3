"""Host-side obstacle fields (curobo_amd/scene/primitives.py): the mesh signed distance equals the
closed-form field of the same solid, primitives obey their defining properties, and a baked ESDF
reproduces the field within the grid's interpolation error through the oracle's voxel lookup."""

import numpy as np
from pytest import raises as pytest_raises

from curobo_amd.scene import bake_esdf, box_mesh, capsule_sdf, cuboid_sdf, cylinder_sdf, mesh_sdf, sphere_sdf, union_sdf


def test_box_mesh_sdf_equals_cuboid_sdf():
    rng = np.random.default_rng(0)
    dims = (0.6, 0.4, 0.25)
    pose = (0.3, -0.2, 0.5, 0.9238795, 0.0, 0.3826834, 0.0)
    v, f = box_mesh(dims)
    p = rng.uniform(-0.5, 0.5, (4000, 3)) + np.array(pose[:3])
    got, want = mesh_sdf(v, f, pose)(p), cuboid_sdf(dims, pose)(p)
    np.testing.assert_allclose(got, want, atol=1e-9)
    assert (want < 0).sum() > 50  # inside samples were tested too
    # points exactly on faces, edges and vertices of the box: distance 0, no sign flips nearby
    on = np.array([[0.3, 0.0, 0.0], [0.3, 0.2, 0.0], [0.3, 0.2, 0.125]])
    local = mesh_sdf(v, f)
    np.testing.assert_allclose(local(on), 0.0, atol=1e-12)
    assert (local(on * 1.001) > 0).all() and (local(on * 0.999) < 0).all()


def test_primitive_fields():
    rng = np.random.default_rng(1)
    p = rng.uniform(-1, 1, (2000, 3))
    s = sphere_sdf(0.3, (0.1, 0.2, -0.1, 1, 0, 0, 0))
    np.testing.assert_allclose(s(p), np.linalg.norm(p - [0.1, 0.2, -0.1], axis=-1) - 0.3)
    # a capsule with base == tip is a sphere; a long thin capsule contains its axis
    c0 = capsule_sdf(0.3, (0, 0, 0), (0, 0, 0), (0.1, 0.2, -0.1, 1, 0, 0, 0))
    np.testing.assert_allclose(c0(p), s(p), atol=1e-12)
    c = capsule_sdf(0.05, (0, 0, 0), (0, 0, 0.5))
    axis = np.stack([np.zeros(11), np.zeros(11), np.linspace(0, 0.5, 11)], -1)
    np.testing.assert_allclose(c(axis), -0.05, atol=1e-12)
    np.testing.assert_allclose(c(np.array([[0.0, 0.0, 0.7]])), 0.15, atol=1e-12)
    # cylinder: radial and axial distances, corner distance
    cy = cylinder_sdf(0.2, 0.6)
    np.testing.assert_allclose(cy(np.array([[0.5, 0, 0], [0, 0, 0.5], [0, 0, 0], [0.5, 0, 0.7]])),
                               [0.3, 0.2, -0.2, np.hypot(0.3, 0.4)], atol=1e-12)
    # every field is 1-Lipschitz (the culling in the kernels relies on it)
    q = p + rng.normal(0, 0.05, p.shape)
    for fld in (s, c, cy, cuboid_sdf((0.3, 0.2, 0.5), (0, 0.1, 0, 0.7071068, 0.7071068, 0, 0)), union_sdf(s, cy)):
        assert (np.abs(fld(p) - fld(q)) <= np.linalg.norm(p - q, axis=-1) + 1e-12).all()


def test_baked_esdf_reproduces_the_field_through_the_oracle(oracle):
    field = union_sdf(sphere_sdf(0.15, (0.4, 0.0, 0.3, 1, 0, 0, 0)), capsule_sdf(0.05, (0, 0, 0), (0, 0, 0.4), (0.2, 0.3, 0.1, 1, 0, 0, 0)))
    vs = 0.02
    grid = bake_esdf(field, (-0.1, -0.3, -0.1), (0.7, 0.6, 0.7), vs)
    assert grid["voxel_features"].dtype == np.float16
    rng = np.random.default_rng(2)
    pts = rng.uniform([0.0, -0.2, 0.0], [0.6, 0.5, 0.6], (3000, 3)).astype(np.float32)
    radius = np.float32(0.03)
    spheres = np.concatenate([pts, np.full((len(pts), 1), radius, np.float32)], -1).reshape(len(pts), 1, 1, 4)
    eta = 0.02
    out = oracle.scene_collision(spheres, grid, weight=1.0, activation_distance=eta)
    d = field(pts.astype(np.float64))
    pen = radius + eta - d
    want = np.where(pen <= 0, 0.0, np.where(pen > eta, pen - 0.5 * eta, 0.5 * pen * pen / eta))
    got = out["distance"].reshape(-1)
    # trilinear interpolation of a 1-Lipschitz field on a 2 cm grid + fp16 storage
    np.testing.assert_allclose(got, want, atol=0.6 * vs)
    assert (want > 0).sum() > 100 and (want == 0).sum() > 100


def test_voxel_coarse_min_is_a_lower_bound_of_every_interpolated_value():
    """curobo_hip_scene.voxel_coarse_min (backends.collision.build_voxel_coarse_min, CPU tensors here): for every
    fine voxel, the coarse cell that contains it holds a value <= the ESDF at every voxel within `dilate` voxels
    (the corners any trilinear sample within (dilate - 1) voxels of it can touch)."""
    import torch

    from curobo_amd.backends.collision import build_voxel_coarse_min

    rng = np.random.default_rng(0)
    nx, ny, nz, block, dilate = 13, 10, 9, 4, 3
    f = rng.normal(size=(nx, ny, nz)).astype(np.float16)
    prm = np.array([[[nx, ny, nz, 0.02]]], np.float32)
    c = build_voxel_coarse_min(torch.as_tensor(f.reshape(1, 1, -1)), prm, block, dilate).numpy()[0, 0]
    cx, cy, cz = -(-nx // block), -(-ny // block), -(-nz // block)
    c = c[: cx * cy * cz].reshape(cx, cy, cz).astype(np.float32)
    ff = f.astype(np.float32)
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                lo = [max(0, v - dilate) for v in (i, j, k)]
                hi = [min(n, v + dilate + 1) for v, n in ((i, nx), (j, ny), (k, nz))]
                assert c[i // block, j // block, k // block] <= ff[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].min()
    # and it is tight: the cell minimum is attained inside its dilated block
    for a in range(cx):
        for b in range(cy):
            for d in range(cz):
                lo = [max(0, v * block - dilate) for v in (a, b, d)]
                hi = [min(n, v * block + block + dilate) for v, n in ((a, nx), (b, ny), (d, nz))]
                assert c[a, b, d] == ff[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].min()


def test_voxel_coarse_min_skips_empty_grid_slots():
    """Environments with different grid counts pad the [E, n] grid table with all-zero slots: those rows keep the
    'never culls' padding (-65504) and the real slots are unchanged (a zero-sized slot used to reach max_pool3d)."""
    import torch

    from curobo_amd.backends.collision import build_voxel_coarse_min

    rng = np.random.default_rng(1)
    nx, ny, nz = 9, 8, 6
    f = np.zeros((2, 2, nx * ny * nz), np.float16)
    f[0, 0] = rng.normal(size=nx * ny * nz).astype(np.float16)
    f[0, 1] = rng.normal(size=nx * ny * nz).astype(np.float16)
    f[1, 0] = rng.normal(size=nx * ny * nz).astype(np.float16)
    prm = np.zeros((2, 2, 4), np.float32)
    prm[0, 0] = prm[0, 1] = prm[1, 0] = (nx, ny, nz, 0.02)  # env 1 has one grid: slot [1, 1] is padding
    c = build_voxel_coarse_min(torch.as_tensor(f), prm, 4, 3).numpy()
    assert (c[1, 1] == np.float16(-65504.0)).all()
    one = build_voxel_coarse_min(torch.as_tensor(f[1:2, 0:1]), prm[1:2, 0:1], 4, 3).numpy()
    np.testing.assert_array_equal(c[1, 0, : one.shape[-1]], one[0, 0])


PRIM_WORLD = [[
    {"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]},
    {"type": "sphere", "radius": 0.18, "pose": [0.45, 0.1, 0.45, 1, 0, 0, 0]},
    {"type": "capsule", "radius": 0.07, "base": [0, 0, 0.0], "tip": [0, 0, 0.5], "pose": [0.1, 0.5, 0.2, 0.9238795, 0.3826834, 0, 0]},
    {"type": "cylinder", "radius": 0.12, "height": 0.6, "pose": [-0.35, -0.35, 0.4, 0.9659258, 0, 0.2588190, 0]},
]]


def test_device_primitive_records_match_the_closed_forms(oracle):
    """Analytic sphere / capsule / cylinder records of the cuboid store (oracle restatement of the device code) vs
    the independent NumPy closed forms of scene/primitives.py: hit set and cost of random probe spheres (weight 1,
    activation 0: cost = penetration depth r - sdf), gradient vs central finite differences."""
    from curobo_amd.scene import capsule_sdf, cuboid_scene_arrays, cylinder_sdf, sphere_sdf

    arrays = cuboid_scene_arrays(PRIM_WORLD)
    assert arrays["cuboid_dims"][0, :, 3].tolist() == [0.0, 1.0, 2.0, 3.0]
    rng = np.random.default_rng(0)
    n = 4000
    pts = rng.uniform([-0.7, -0.7, 0.05], [0.8, 0.9, 0.9], size=(n, 3))
    rad = rng.uniform(0.01, 0.06, size=n)
    sph = np.concatenate([pts, rad[:, None]], -1).astype(np.float32).reshape(1, 1, n, 4)
    fields = [sphere_sdf(0.18, PRIM_WORLD[0][1]["pose"]),
              capsule_sdf(0.07, [0, 0, 0.0], [0, 0, 0.5], PRIM_WORLD[0][2]["pose"]),
              cylinder_sdf(0.12, 0.6, PRIM_WORLD[0][3]["pose"])]
    for k, f in enumerate(fields):
        only = cuboid_scene_arrays([[dict(o, enable=(i == k + 1)) for i, o in enumerate(PRIM_WORLD[0])]])
        r = oracle.scene_collision(sph, only, 1.0, 0.0)
        sdf = f(pts.astype(np.float32).astype(np.float64))
        pen = rad.astype(np.float32).astype(np.float64) - sdf
        clear = np.abs(pen) > 1e-5
        assert np.array_equal((r["distance"][0, 0] > 0)[clear], (pen > 0)[clear]), k
        hit = (pen > 1e-4)
        assert hit.sum() > 20
        np.testing.assert_allclose(r["distance"][0, 0][hit], pen[hit], rtol=2e-4, atol=2e-6)
        # gradient of the cost wrt the sphere centre = -grad sdf: central differences of the closed form
        idx = np.where(hit)[0][:64]
        eps = 1e-4
        for a in range(3):
            dp = np.zeros(3); dp[a] = eps
            fd = -(f(pts[idx].astype(np.float32) + dp) - f(pts[idx].astype(np.float32) - dp)) / (2 * eps)
            np.testing.assert_allclose(r["gradient"][0, 0][idx, a], fd, atol=2e-2)


def test_primitive_world_costs_add_up(oracle):
    """all four records together = the sum of the single-record scenes (obstacle-index order)"""
    from curobo_amd.scene import cuboid_scene_arrays

    rng = np.random.default_rng(1)
    sph = np.concatenate([rng.uniform([-0.7, -0.7, 0.0], [0.8, 0.9, 0.9], size=(2, 5, 40, 3)), rng.uniform(0.02, 0.08, size=(2, 5, 40, 1))],
                         -1).astype(np.float32)
    full = oracle.scene_collision(sph, cuboid_scene_arrays(PRIM_WORLD), 3.0, 0.01, sweep=True)
    parts = [oracle.scene_collision(sph, cuboid_scene_arrays([[dict(o, enable=(i == k)) for i, o in enumerate(PRIM_WORLD[0])]]), 3.0, 0.01,
                                    sweep=True) for k in range(4)]
    np.testing.assert_allclose(full["distance"], sum(p["distance"] for p in parts), rtol=1e-5, atol=1e-6)
    assert all((p["distance"] > 0).any() for p in parts)


def test_obstacle_transform_matrix_and_bounding_sphere():
    """``Obstacle.get_transform_matrix`` / ``get_sphere`` (reference geom/types.py:160-194)"""
    from scipy.spatial.transform import Rotation

    from curobo_amd.scene.types import Cuboid, Cylinder

    pose = [0.1, 0.2, 0.3, 0.924, 0.0, 0.383, 0.0]
    c = Cuboid("a", pose, dims=[0.3, 0.2, 0.1])
    m = c.get_transform_matrix()
    q = np.asarray(pose[3:]) / np.linalg.norm(pose[3:])
    np.testing.assert_allclose(m[:3, :3], Rotation.from_quat(q[[1, 2, 3, 0]]).as_matrix(), atol=1e-12)
    np.testing.assert_allclose(m[:, 3], [0.1, 0.2, 0.3, 1.0])
    s = c.get_sphere()
    assert s.radius == 0.1 and list(s.pose) == pose  # (the cuboid's smallest edge, as the reference takes it)
    assert abs(Cylinder("c", pose=[0, 0, 1, 1, 0, 0, 0], radius=0.1, height=0.5).get_sphere().radius - 0.2) < 1e-12


def test_stl_files_binary_and_ascii(tmp_path):
    """``scene.mesh.load_stl`` / ``load_mesh_file``: a cube written as binary STL (with a header that starts with "solid") and as ASCII STL
    reads back as 8 welded vertices and 12 faces with the volume of the cube; the reference's own STL assets load when they are here"""
    import os
    import struct

    from curobo_amd.scene.mesh import load_mesh_file, load_stl
    from curobo_amd.scene.types import Mesh

    c = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32) * 0.2
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = np.array([t for q in quads for t in ([c[q[0]], c[q[1]], c[q[2]]], [c[q[0]], c[q[2]], c[q[3]]])], np.float32)
    binary = tmp_path / "cube.stl"
    with open(binary, "wb") as fh:
        fh.write(b"solid binary files may begin like this".ljust(80, b" "))
        fh.write(struct.pack("<I", len(tris)))
        for t in tris:
            fh.write(struct.pack("<12fH", 0, 0, 0, *t.reshape(-1), 0))
    ascii_ = tmp_path / "cube_ascii.STL"
    with open(ascii_, "w") as fh:
        fh.write("solid cube\n")
        for t in tris:
            fh.write(" facet normal 0 0 0\n  outer loop\n" + "".join(f"   vertex {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n" for p in t) + "  endloop\n endfacet\n")
        fh.write("endsolid cube\n")
    for path in (binary, ascii_):
        v, f = load_mesh_file(str(path))
        assert v.shape == (8, 3) and f.shape == (12, 3)
        a, b, d = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
        assert abs(abs(np.einsum("ij,ij->i", a, np.cross(b, d)).sum() / 6.0) - 0.2 ** 3) < 1e-9  # (divergence theorem: consistent winding)
    mv, mf = Mesh(name="m", pose=[0, 0, 0, 1, 0, 0, 0], file_path=str(binary)).get_mesh_data()
    assert len(mv) == 8 and len(mf) == 12
    with pytest_raises(ValueError):
        load_mesh_file(str(tmp_path / "thing.dae"))
    ref = "/root/reference/curobo/content/assets/robot"
    if os.path.isdir(ref):
        found = [os.path.join(d, n) for d, _, names in os.walk(ref) for n in names if n.lower().endswith(".stl")][:5]
        for p in found:
            v, f = load_stl(p)
            assert v.shape[0] >= 4 and f.shape[0] >= 4 and int(f.max()) < v.shape[0]
