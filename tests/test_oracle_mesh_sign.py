"""The SIGN of a mesh query, pinned to the reference's rule (row f3 of SURVEY section 8).

The reference asks NVIDIA Warp: ``wp.mesh_query_point(mesh, point, max_distance)`` (``curobo/_src/geom/data/data_mesh.py:632,
682``), whose published algorithm (warp-lang >= 0.10.0 per the reference's pyproject.toml:37; ``warp/native/mesh.h``,
``mesh_query_point`` -> ``mesh_query_inside``) decides inside / outside with THREE RAYS from the query point along +x, +y, +z:
each ray's nearest hit says whether it met the front or the back of a face, and the point is inside iff all three rays hit and all
three hits are back faces.  Warp is not in /root/reference and cannot be imported here; the oracle restates that rule in double
precision over every triangle (``orc_mesh_inside_rays``, rule "rays").  This file

1. holds the three rules in play -- Warp's rays, the winding number (the oracle's default since round 3) and the closest-feature
   pseudonormal rule the HIP kernels use on closed meshes -- to ONE function on closed, consistently oriented meshes (box, thin
   plate, concave L, sphere, torus, at random poses), which is why the kernels may use the cheap one there;
2. states where they PART on meshes that are not closed or not consistently oriented (an open box, a single-sided plate, a box
   with one flipped face), with the reference's answers written out as facts of its rule, and holds the product's switch:
   ``mesh_is_closed_and_oriented`` is False exactly on those fixtures, and such a mesh is signed with Warp's rays on the device
   (``SIGN_WARP_RAYS``; GPU side: ``tests/test_gpu_mesh.py::test_open_and_flipped_meshes_take_the_reference_ray_sign``);
3. mirrors the reference's one mesh test (``tests/_src/collision/test_mesh_collision_sdf.py``: a 5 cm cube as mesh and as
   cuboid, four probe distances) under the ray rule -- it is in ``test_oracle_mesh.py`` under the winding rule.
"""
import numpy as np
import pytest

from test_oracle_mesh import box_shape, ell_shape, small_cube_case, sphere_shape, subdivide, torus_shape
from mesh_sign_rules import pseudonormal_sign


# ----------------------------------------------------------------------------------------------- fixtures
def rotated(v, seed):
    """a generic rigid rotation (no face normal or edge stays axis aligned: the +x / +y / +z rays then meet no edge exactly)"""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return (np.asarray(v, np.float64) @ q.T).astype(np.float32), q


def face_normals(v, f):
    a, b, c = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
    n = np.cross(b - a, c - a)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def without_faces_facing(v, f, direction):
    """the mesh with the faces whose outward normal is `direction` removed (an open box)"""
    keep = face_normals(v, f) @ np.asarray(direction, np.float64) < 0.999
    return v, f[keep]


def with_flipped(v, f, which):
    f = f.copy()
    f[which] = f[which][:, [0, 2, 1]]
    return v, f


def plate(nx=6, ny=6, size=0.4):
    """a single-sided square sheet in the plane z = 0, normals +z (an OPEN mesh: one face of a thin plate)"""
    xs, ys = np.linspace(-size / 2, size / 2, nx + 1), np.linspace(-size / 2, size / 2, ny + 1)
    v = np.array([[x, y, 0.0] for x in xs for y in ys], np.float32)
    idx = lambda i, j: i * (ny + 1) + j  # noqa: E731
    f = []
    for i in range(nx):
        for j in range(ny):
            f += [[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)], [idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)]]
    return v, np.asarray(f, np.int32)


CLOSED = {
    "box": lambda: box_shape([0.3, 0.5, 0.2], 2),
    "thin_plate": lambda: box_shape([0.4, 0.4, 0.002], 3),   # a closed plate 2 mm thick
    "concave_L": lambda: ell_shape(2),
    "sphere": lambda: sphere_shape(0.2, 12, 24),
    "torus": lambda: torus_shape(0.22, 0.06, 24, 12),
}


def probes(v, n, seed, margin=0.15):
    rng = np.random.default_rng(seed)
    lo, hi = v.min(0) - margin, v.max(0) + margin
    return rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)


def signs(oracle, rule, p, v, f):
    oracle.set_mesh_sign_rule(rule)
    try:
        sdf, _ = oracle.mesh_query(p, v, f, 100.0)
    finally:
        oracle.set_mesh_sign_rule("winding")
    return np.sign(sdf).astype(int), np.abs(sdf)


# ----------------------------------------------------------------------------------------------- 1. closed meshes: one function
@pytest.mark.parametrize("name", sorted(CLOSED))
def test_closed_meshes_three_rules_one_function(oracle, name):
    from curobo_amd.backends.mesh import mesh_is_closed_and_oriented

    v0, f = CLOSED[name]()
    assert mesh_is_closed_and_oriented(v0, f)
    total_in = 0
    for seed in (0, 1):
        v, _ = rotated(v0, 10 + seed)
        p = probes(v, 1500, seed)
        s_ray, d = signs(oracle, "rays", p, v, f)
        s_wn, d2 = signs(oracle, "winding", p, v, f)
        s_pn, d3 = pseudonormal_sign(p, v, f)
        np.testing.assert_array_equal(d, d2)              # the distance is the same brute force under both rules
        np.testing.assert_allclose(d3, d, atol=2e-6)
        off = d > 1e-5                                    # on the surface itself the sign is rounding under every rule
        np.testing.assert_array_equal(s_ray[off], s_wn[off])
        np.testing.assert_array_equal(s_pn[off], s_ray[off])
        total_in += int((s_ray[off] < 0).sum())
    assert total_in > (3 if name == "thin_plate" else 40)  # the probes do reach the inside


def test_unrotated_closed_box_under_the_ray_rule(oracle):
    """axis-aligned faces, generic probe points: the rays hit face interiors; same function again"""
    v, f = box_shape([0.3, 0.5, 0.2], 2)
    p = probes(v, 2000, 5)
    s_ray, d = signs(oracle, "rays", p, v, f)
    s_wn, _ = signs(oracle, "winding", p, v, f)
    off = d > 1e-5
    np.testing.assert_array_equal(s_ray[off], s_wn[off])
    inside_box = (np.abs(p) < np.array([0.15, 0.25, 0.1]) - 1e-4).all(1)
    assert inside_box.sum() > 50 and (s_ray[inside_box] < 0).all()


# ----------------------------------------------------------------------------------------------- 2. where the rules part
def test_single_sided_plate_reference_rule_is_unsigned(oracle):
    """a sheet with normals +z: the +x and +y rays of a generic point never meet it, so Warp's vote is never 3 -- the reference
    reports a POSITIVE distance on both sides (an unsigned distance field); the closest-feature rule calls everything below the
    sheet 'inside' (penetration growing with depth), the winding number calls nothing inside.  Not closed -> the product casts
    the rays."""
    from curobo_amd.backends.mesh import mesh_is_closed_and_oriented

    v, f = plate()
    assert not mesh_is_closed_and_oriented(v, f)
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.uniform(-0.15, 0.15, size=(400, 2)), rng.uniform(-0.2, 0.2, size=(400, 1))], 1).astype(np.float32)
    p = p[np.abs(p[:, 2]) > 1e-3]
    s_ray, d = signs(oracle, "rays", p, v, f)
    assert (s_ray > 0).all()
    np.testing.assert_allclose(d, np.abs(p[:, 2]), atol=1e-6)
    s_pn, _ = pseudonormal_sign(p, v, f)
    below = p[:, 2] < 0
    assert (s_pn[below] < 0).all() and (s_pn[~below] > 0).all()          # the rule NOT to use here
    s_wn, _ = signs(oracle, "winding", p, v, f)
    assert (s_wn > 0).all()


def test_open_box_depends_on_which_face_is_missing(oracle):
    """An asymmetry of the reference's rule worth knowing: the rays go along +x, +y, +z only.  A box that lacks its +z face has no
    inside at all (the +z ray of an interior point escapes); a box that lacks its -z face still has its whole interior inside (no
    ray goes that way)."""
    from curobo_amd.backends.mesh import mesh_is_closed_and_oriented

    dims = np.array([0.3, 0.5, 0.2])
    v, f = box_shape(dims.tolist(), 2)
    rng = np.random.default_rng(1)
    inner = (rng.uniform(-0.5, 0.5, size=(300, 3)) * (dims - 2e-3)).astype(np.float32)
    outer = probes(v, 600, 2)
    outer = outer[(np.abs(outer) > dims / 2 + 1e-3).any(1)]
    for direction, interior_is_inside in (([0, 0, 1], False), ([0, 0, -1], True), ([1, 0, 0], False), ([-1, 0, 0], True)):
        vo, fo = without_faces_facing(v, f, direction)
        assert len(fo) < len(f) and not mesh_is_closed_and_oriented(vo, fo)
        s_in, _ = signs(oracle, "rays", inner, vo, fo)
        assert ((s_in < 0) == interior_is_inside).all(), direction
        s_out, _ = signs(oracle, "rays", outer, vo, fo)
        assert (s_out > 0).all(), direction                # outside stays outside whichever face is missing
    # the closest-feature rule on the same open box: the interior stays inside whichever face is missing (every remaining face
    # still turns its back to it) -- the two rules differ on the WHOLE interior of a box open towards +x, +y or +z
    vo, fo = without_faces_facing(v, f, [0, 0, 1])
    deep = inner[np.abs(inner[:, 2]) < 0.05]
    s_pn, _ = pseudonormal_sign(deep, vo, fo)
    assert (s_pn < 0).mean() > 0.9


def test_one_flipped_face_shadows_along_the_ray_axes(oracle):
    """a closed box with one triangle of its +x wall wound the other way: interior points whose +x ray leaves through that triangle
    meet a FRONT face there and are reported outside; every other interior point is unaffected; the closest-feature rule instead
    flips the points whose CLOSEST feature is that triangle, on both sides of the wall"""
    from curobo_amd.backends.mesh import mesh_is_closed_and_oriented

    dims = np.array([0.3, 0.5, 0.2])
    v, f = box_shape(dims.tolist(), 1)
    n = face_normals(v, f)
    wall = np.flatnonzero(n[:, 0] > 0.999)
    t = wall[len(wall) // 2]
    vf, ff = with_flipped(v, f, [t])
    assert not mesh_is_closed_and_oriented(vf, ff)
    a, b, c = (v[f[t, k]].astype(np.float64) for k in range(3))
    rng = np.random.default_rng(3)
    inner = (rng.uniform(-0.5, 0.5, size=(4000, 3)) * (dims - 2e-3)).astype(np.float32)
    # barycentric test of the point's (y, z) against the flipped triangle's (y, z): does the +x ray leave through it?
    def in_tri(p):
        m = np.array([[b[1] - a[1], c[1] - a[1]], [b[2] - a[2], c[2] - a[2]]])
        uv = np.linalg.solve(m, (p[:, 1:3].astype(np.float64) - a[1:3]).T).T
        return (uv[:, 0] > 1e-3) & (uv[:, 1] > 1e-3) & (uv.sum(1) < 1 - 1e-3), (uv[:, 0] < -1e-3) | (uv[:, 1] < -1e-3) | (uv.sum(1) > 1 + 1e-3)
    shadow, clear = in_tri(inner)
    assert shadow.sum() > 20
    s_ray, _ = signs(oracle, "rays", inner, vf, ff)
    assert (s_ray[shadow] > 0).all() and (s_ray[clear] < 0).all()
    # outside, just beyond the wall in front of the flipped triangle: the reference still says outside (the +x ray hits nothing)
    front = np.stack([np.full(50, dims[0] / 2 + 0.01), rng.uniform(-0.2, 0.2, 50), rng.uniform(-0.08, 0.08, 50)], 1).astype(np.float32)
    s_front, _ = signs(oracle, "rays", front, vf, ff)
    assert (s_front > 0).all()
    near = front[in_tri(front)[0]]
    if len(near):
        s_pn, _ = pseudonormal_sign(near, vf, ff)
        assert (s_pn < 0).all()                             # the closest feature is the flipped face: 'inside' by that rule


def test_disagreement_table_over_the_fixtures(oracle):
    """the numbers DESIGN section 4.3 quotes: share of off-surface probes on which each cheap rule differs from the reference's"""
    rows = {}
    fixtures = {k: fn() for k, fn in CLOSED.items()}
    vb, fb = box_shape([0.3, 0.5, 0.2], 2)
    fixtures["open_box(+z missing)"] = without_faces_facing(vb, fb, [0, 0, 1])
    fixtures["open_box(-z missing)"] = without_faces_facing(vb, fb, [0, 0, -1])
    fixtures["single_sided_plate"] = plate()
    n = face_normals(vb, fb)
    fixtures["box_one_flipped_face"] = with_flipped(vb, fb, [np.flatnonzero(n[:, 0] > 0.999)[0]])
    for name, (v, f) in fixtures.items():
        p = probes(v, 1200, 7, margin=0.1)
        s_ray, d = signs(oracle, "rays", p, v, f)
        s_wn, _ = signs(oracle, "winding", p, v, f)
        s_pn, _ = pseudonormal_sign(p, v, f)
        off = d > 1e-5
        rows[name] = (float((s_pn[off] != s_ray[off]).mean()), float((s_wn[off] != s_ray[off]).mean()))
    for name in CLOSED:
        assert rows[name] == (0.0, 0.0), (name, rows[name])
    assert rows["open_box(+z missing)"][0] > 0.05            # the interior: inside by the closest feature, outside by the rays
    assert rows["single_sided_plate"][0] > 0.2
    assert rows["box_one_flipped_face"][0] > 0.0
    print("\nshare of probes where (pseudonormal, winding) differ from Warp's rays:", {k: (round(a, 3), round(b, 3)) for k, (a, b) in rows.items()})


# ----------------------------------------------------------------------------------------------- 3. the reference's own test
def test_reference_regression_small_mesh_cost_matches_cuboid_under_the_ray_rule(oracle):
    """tests/_src/collision/test_mesh_collision_sdf.py:17-60 case by case: a 5 cm cube as mesh and as cuboid, probe spheres of
    radius 5 cm at 0.08 / 0.10 / 0.50 / 1.00 m, weight 1, activation 0.01: the two costs agree (torch.allclose defaults), the
    nearest probe collides, the other three do not"""
    from oracle.oracle import mesh_scene_arrays

    from curobo_amd.scene import cuboid_scene_arrays

    v, f, sph = small_cube_case()
    pose = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    oracle.set_mesh_sign_rule("rays")
    try:
        mesh_cost = oracle.scene_collision(sph, mesh_scene_arrays([[{"name": "box", "vertices": v, "faces": f, "pose": pose}]]), 1.0, 0.01)["distance"]
    finally:
        oracle.set_mesh_sign_rule("winding")
    cub_cost = oracle.scene_collision(sph, cuboid_scene_arrays([[{"dims": [0.05] * 3, "pose": pose}]]), 1.0, 0.01)["distance"]
    np.testing.assert_allclose(mesh_cost, cub_cost, rtol=1e-5, atol=1e-8)
    assert mesh_cost.reshape(-1)[0] > 0.0 and (mesh_cost.reshape(-1)[1:] == 0.0).all()
