"""Mesh obstacles in the oracle: brute-force closest triangle + winding-number sign (oracle/curobo_oracle.c orc_mesh_sdf_raw)
against closed forms of the shapes the meshes represent, and through the scene-collision restatement.

What is pinned and what is not: the reference reaches meshes through NVIDIA Warp's ``wp.mesh_query_point`` (data_mesh.py:630-700);
Warp is outside /root/reference and has no ROCm build, so its BVH walk cannot be run here.  The contract the reference relies
on is pinned instead -- signed distance negative inside, gradient (p - closest) / |p - closest|, ``max_distance`` when nothing
lies within ``max(half bounding-box diagonal, query distance)`` -- on shapes whose exact signed distance is known in closed form.
"""

import numpy as np
import pytest


# ----------------------------------------------------------------------------------------------- shapes (shared with the GPU tests)
def subdivide(v, f, times=1):
    """each triangle -> 4 (edge midpoints); the surface is unchanged, the BVH gets something to do"""
    v = [tuple(x) for x in np.asarray(v, np.float64)]
    f = np.asarray(f, np.int64)
    for _ in range(times):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                cache[key] = len(v)
                v.append(tuple((np.asarray(v[a]) + np.asarray(v[b])) * 0.5))
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
        f = np.asarray(out, np.int64)
    return np.asarray(v, np.float32), f.astype(np.int32)


def box_shape(dims, times=2):
    from curobo_amd.scene import box_mesh

    return subdivide(*box_mesh(dims), times=times)


def sphere_shape(radius, n_lat=24, n_lon=48):
    v = [[0, 0, radius]]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            v.append([radius * np.sin(th) * np.cos(ph), radius * np.sin(th) * np.sin(ph), radius * np.cos(th)])
    v.append([0, 0, -radius])
    f = []
    ring = lambda i, j: 1 + (i - 1) * n_lon + j % n_lon  # noqa: E731
    for j in range(n_lon):
        f.append([0, ring(1, j), ring(1, j + 1)])
        f.append([len(v) - 1, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)])
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            f.append([ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)])
            f.append([ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def torus_shape(R, r, n_major=48, n_minor=24):
    """non-convex, genus 1: a point on the axis is outside although every ray from it hits the surface"""
    v, f = [], []
    for i in range(n_major):
        a = 2 * np.pi * i / n_major
        for j in range(n_minor):
            b = 2 * np.pi * j / n_minor
            v.append([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)])
    idx = lambda i, j: (i % n_major) * n_minor + j % n_minor  # noqa: E731
    for i in range(n_major):
        for j in range(n_minor):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def torus_sdf(R, r):
    return lambda p: np.sqrt((np.sqrt(p[..., 0] ** 2 + p[..., 1] ** 2) - R) ** 2 + p[..., 2] ** 2) - r


def ell_shape(times=2):
    """an L-shaped prism (non-convex polyhedron, exact): [0, .4]x[0, .4]x[0, .2] minus (.2, .4]x(.2, .4]x[0, .2]"""
    xy = np.array([[0, 0], [0.4, 0], [0.4, 0.2], [0.2, 0.2], [0.2, 0.4], [0, 0.4]], np.float64)
    n = len(xy)
    v = [[x, y, 0.0] for x, y in xy] + [[x, y, 0.2] for x, y in xy]
    f = []
    for i in range(n):  # walls, outward normals for the counter-clockwise outline
        j = (i + 1) % n
        f += [[i, j, n + j], [i, n + j, n + i]]
    for a, b, c in [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5)]:  # fan from the corner that sees the whole outline
        f.append([a, c, b])              # bottom (normal -z)
        f.append([n + a, n + b, n + c])  # top (normal +z)
    return subdivide(np.asarray(v), np.asarray(f), times=times)


def ell_sdf_outside(p):
    """exact for points outside the prism: the minimum over the two boxes whose union it is"""
    from curobo_amd.scene import cuboid_sdf, union_sdf

    return union_sdf(cuboid_sdf([0.4, 0.2, 0.2], [0.2, 0.1, 0.1, 1, 0, 0, 0]), cuboid_sdf([0.2, 0.4, 0.2], [0.1, 0.2, 0.1, 1, 0, 0, 0]))(p)


def is_closed_and_oriented(f):
    """every directed edge appears once and its reverse once"""
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    fwd = {(int(a), int(b)) for a, b in e}
    return len(fwd) == len(e) and all((b, a) in fwd for a, b in fwd)


#: mesh world of the collision tests: a table as a cuboid mesh and the three shapes of PRIM_WORLD tessellated, + a torus
def mesh_world(pose_jitter=0.0):
    vb, fb = box_shape([2.2, 2.2, 0.2], 3)
    vs, fs = sphere_shape(0.18)
    vt, ft = torus_shape(0.22, 0.06)
    ve, fe = ell_shape()
    return [[
        {"name": "table", "vertices": vb, "faces": fb, "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]},
        {"name": "ball", "vertices": vs, "faces": fs, "pose": [0.45, 0.1, 0.45, 0.9238795, 0, 0.3826834, 0]},
        {"name": "ring", "vertices": vt, "faces": ft, "pose": [0.1, 0.45, 0.45, 0.9238795, 0.3826834, 0, 0]},
        {"name": "ell", "vertices": ve, "faces": fe, "pose": [-0.45, -0.45, 0.3, 0.9659258, 0, 0.2588190, 0]},
        {"name": "ball2", "mesh_name": "ball", "vertices": vs, "faces": fs, "pose": [-0.4, 0.3, 0.7, 1, 0, 0, 0], "enable": False},
    ]]


# ----------------------------------------------------------------------------------------------- tests
def test_shapes_are_closed_and_consistently_oriented():
    for v, f in (box_shape([0.3, 0.5, 0.2]), sphere_shape(0.2), torus_shape(0.22, 0.06), ell_shape()):
        assert is_closed_and_oriented(f)
        # outward orientation: positive signed volume
        a, b, c = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
        assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0


def test_box_mesh_is_the_cuboid_signed_distance(oracle):
    from curobo_amd.scene import cuboid_sdf

    dims = [0.3, 0.5, 0.2]
    v, f = box_shape(dims)
    rng = np.random.default_rng(0)
    p = rng.uniform(-0.45, 0.45, size=(3000, 3)).astype(np.float32)
    sdf, grad = oracle.mesh_query(p, v, f, 10.0)
    want = cuboid_sdf(dims)(p.astype(np.float64))
    np.testing.assert_allclose(sdf, want, atol=2e-6)
    assert (sdf < 0).sum() > 100 and (sdf > 0).sum() > 1000
    # gradient: unit vector from the closest surface point to the query (data_mesh.py:693-697) -> outside it is the gradient
    # of the distance field, inside its negative
    eps = 1e-4
    clear = np.abs(sdf) > 5e-3
    fd = np.stack([(cuboid_sdf(dims)(p + np.eye(3)[a] * eps) - cuboid_sdf(dims)(p - np.eye(3)[a] * eps)) / (2 * eps) for a in range(3)], -1)
    smooth = clear & (np.abs(np.linalg.norm(fd, axis=-1) - 1) < 1e-3)  # away from the medial axis
    np.testing.assert_allclose(grad[smooth], (np.sign(sdf)[:, None] * fd)[smooth], atol=5e-3)
    np.testing.assert_allclose(np.linalg.norm(grad[clear], axis=-1), 1.0, atol=1e-5)


def test_sphere_and_torus_meshes_follow_their_closed_forms(oracle):
    rng = np.random.default_rng(1)
    p = rng.uniform(-0.4, 0.4, size=(3000, 3)).astype(np.float32)
    v, f = sphere_shape(0.2, 48, 96)
    sdf, _ = oracle.mesh_query(p, v, f, 10.0)
    want = np.linalg.norm(p, axis=-1) - 0.2
    np.testing.assert_allclose(sdf, want, atol=3e-4)  # chord error of the tessellation: r (1 - cos(pi / 96)) = 1.1e-4
    far = np.abs(want) > 1e-3
    assert np.array_equal(sdf[far] < 0, want[far] < 0)
    v, f = torus_shape(0.22, 0.06, 96, 48)
    sdf, _ = oracle.mesh_query(p, v, f, 10.0)
    want = torus_sdf(0.22, 0.06)(p.astype(np.float64))
    np.testing.assert_allclose(sdf, want, atol=3e-4)
    far = np.abs(want) > 1e-3
    assert np.array_equal(sdf[far] < 0, want[far] < 0)
    assert (sdf < 0).sum() > 30
    # on the axis (inside the hole): outside, although surface surrounds the point
    s0, _ = oracle.mesh_query(np.zeros((1, 3), np.float32), v, f, 10.0)
    assert abs(s0[0] - 0.16) < 3e-4


def test_nonconvex_prism_sign_and_exterior_distance(oracle):
    v, f = ell_shape()
    rng = np.random.default_rng(2)
    p = rng.uniform(-0.15, 0.55, size=(4000, 3)).astype(np.float32)
    p[:, 2] = rng.uniform(-0.15, 0.35, size=4000)
    sdf, _ = oracle.mesh_query(p, v, f, 10.0)
    want = ell_sdf_outside(p.astype(np.float64))
    clear = np.abs(want) > 1e-5
    assert np.array_equal((sdf < 0)[clear], (want < 0)[clear])
    out = want > 0
    np.testing.assert_allclose(sdf[out], want[out], atol=2e-6)
    notch = (p[:, 0] > 0.25) & (p[:, 1] > 0.25) & (p[:, 2] > 0.05) & (p[:, 2] < 0.15)  # the removed corner: outside
    assert notch.sum() > 20 and (sdf[notch] > 0).all()


def test_max_distance_cutoff(oracle):
    v, f = box_shape([0.2, 0.2, 0.2])
    p = np.array([[0.5, 0, 0], [0.15, 0, 0], [0, 0, 0]], np.float32)
    sdf, grad = oracle.mesh_query(p, v, f, 0.1)
    assert sdf[0] == np.float32(0.1) and not grad[0].any()      # nothing within 0.1: (max_distance, 0) (data_mesh.py:682-683)
    assert abs(sdf[1] - 0.05) < 1e-6 and abs(sdf[2] + 0.1) < 1e-6


def test_mesh_world_through_scene_collision_matches_analytic_records(oracle):
    """the mesh kind of the scene restatement: a sphere and a cuboid as meshes vs the same obstacles as analytic records of the
    cuboid store (tessellation error only), discrete and swept; a disabled slot contributes nothing; costs of kinds add"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.scene import cuboid_scene_arrays

    vs, fs = sphere_shape(0.18, 48, 96)
    vb, fb = box_shape([0.3, 0.4, 0.5])
    c, s = np.cos(0.4), np.sin(0.4)
    poses = [[0.45, 0.1, 0.45, 1, 0, 0, 0], [0.3, 0.5, 0.4, c, 0, 0, s]]
    meshes = mesh_scene_arrays([[{"name": "ball", "vertices": vs, "faces": fs, "pose": poses[0]},
                                 {"name": "box", "vertices": vb, "faces": fb, "pose": poses[1]},
                                 {"name": "off", "mesh_name": "box", "vertices": vb, "faces": fb, "pose": [0, 0, 0.5, 1, 0, 0, 0], "enable": False}]])
    analytic = cuboid_scene_arrays([[{"type": "sphere", "radius": 0.18, "pose": poses[0]}, {"dims": [0.3, 0.4, 0.5], "pose": poses[1]}]])
    np.testing.assert_allclose(meshes["mesh_dims"][0, 1, :3], [0.3, 0.4, 0.5], atol=1e-6)
    rng = np.random.default_rng(3)
    b, h, S = 6, 7, 30
    start = rng.uniform([-0.1, -0.2, 0.1], [0.8, 0.9, 0.8], size=(b, 1, S, 3))
    step = rng.normal(size=(b, 1, S, 3)) * 0.02
    pos = start + step * np.arange(h)[None, :, None, None]
    sph = np.concatenate([pos, np.broadcast_to(rng.uniform(0.02, 0.06, size=(1, 1, S, 1)), (b, h, S, 1))], -1).astype(np.float32)
    for sweep in (False, True):
        rm = oracle.scene_collision(sph, meshes, 3.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
        ra = oracle.scene_collision(sph, analytic, 3.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
        assert (ra["distance"] > 0).mean() > 0.05
        np.testing.assert_allclose(rm["distance"], ra["distance"], atol=2e-3, rtol=2e-2)
        both = {**analytic, **meshes}
        rb = oracle.scene_collision(sph, both, 3.0, 0.02, sweep=sweep, enable_speed_metric=sweep, speed_dt=0.05)
        if not sweep:  # (the speed metric scales each kind's share by the same factor only in the discrete case)
            np.testing.assert_allclose(rb["distance"], rm["distance"] + ra["distance"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rb["gradient"], rm["gradient"] + ra["gradient"], rtol=1e-4, atol=1e-5)


def small_cube_case():
    """the inputs of the reference's own regression test, tests/_src/collision/test_mesh_collision_sdf.py:17-60: a 5 cm cube as
    a mesh and as a cuboid, probe spheres of radius 5 cm at 0.08, 0.10, 0.50, 1.00 m, weight 1, activation 0.01"""
    from curobo_amd.scene import box_mesh

    v, f = box_mesh([0.05, 0.05, 0.05])
    sph = np.array([[[[d, 0.0, 0.0, 0.05] for d in (0.08, 0.10, 0.50, 1.00)]]], np.float32)
    return v, f, sph


def test_reference_regression_small_mesh_cost_matches_cuboid(oracle):
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.scene import cuboid_scene_arrays

    v, f, sph = small_cube_case()
    pose = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    mesh_cost = oracle.scene_collision(sph, mesh_scene_arrays([[{"name": "box", "vertices": v, "faces": f, "pose": pose}]]), 1.0, 0.01)["distance"]
    cub_cost = oracle.scene_collision(sph, cuboid_scene_arrays([[{"dims": [0.05] * 3, "pose": pose}]]), 1.0, 0.01)["distance"]
    np.testing.assert_allclose(mesh_cost, cub_cost, rtol=1e-5, atol=1e-8)  # torch.allclose defaults
    assert mesh_cost.reshape(-1)[0] > 0.0 and (mesh_cost.reshape(-1)[1:] == 0.0).all()
