"""Brute-force statements of the sign rules a mesh query can use, for the tests (numpy, fp64; small meshes only):

* ``closest_feature(p, v, f)``: per point the closest triangle, the region of the triangle its closest point lies in (Ericson's
  order, the numbering of ``csrc/mesh_device.hpp::closest_on_triangle``) and the closest point;
* ``pseudonormal_sign``: the rule of the HIP path on closed meshes -- sign of (p - closest) . n of the closest FEATURE, n from the
  product's own table (``curobo_amd.backends.mesh.feature_pseudonormals``);
* the reference's rule (Warp's three axis rays) and the winding number live in the C oracle (``Oracle.set_mesh_sign_rule``).
"""
import numpy as np


def closest_feature(p, v, f):
    p = np.asarray(p, np.float64)[:, None, :]
    a, b, c = (np.asarray(v, np.float64)[np.asarray(f)[:, k]][None] for k in range(3))
    ab, ac, ap = b - a, c - a, p - a
    dot = lambda x, y: np.einsum("...i,...i->...", x, y)  # noqa: E731
    d1, d2 = dot(ab, ap), dot(ac, ap)
    bp = p - b
    d3, d4 = dot(ab, bp), dot(ac, bp)
    cp = p - c
    d5, d6 = dot(ab, cp), dot(ac, cp)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    e43, e56 = d4 - d3, d5 - d6
    at_a = (d1 <= 0) & (d2 <= 0)
    at_b = (d3 >= 0) & (d4 <= d3)
    on_ab = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    at_c = (d6 >= 0) & (d5 <= d6)
    on_ca = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    on_bc = (va <= 0) & (e43 >= 0) & (e56 >= 0)
    region = np.select([at_a, at_b, on_ab, at_c, on_ca, on_bc], [1, 2, 4, 3, 6, 5], 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        q_ab = a + (d1 / (d1 - d3))[..., None] * ab
        q_ca = a + (d2 / (d2 - d6))[..., None] * ac
        q_bc = b + (e43 / (e43 + e56))[..., None] * (c - b)
        den = 1.0 / (va + vb + vc)
        q_f = a + (vb * den)[..., None] * ab + (vc * den)[..., None] * ac
    q = np.select([(region == 1)[..., None], (region == 2)[..., None], (region == 3)[..., None], (region == 4)[..., None],
                   (region == 5)[..., None], (region == 6)[..., None]], [a + 0 * p, b + 0 * p, c + 0 * p, q_ab, q_bc, q_ca], q_f)
    d2all = ((p - q) ** 2).sum(-1)
    t = d2all.argmin(1)
    i = np.arange(p.shape[0])
    return t, region[i, t], q[i, t], np.sqrt(d2all[i, t])


def pseudonormal_sign(p, v, f):
    """+1 outside / -1 inside / 0 no verdict, by the closest feature's (pseudo)normal; and the distance"""
    from curobo_amd.backends.mesh import feature_pseudonormals

    pn = feature_pseudonormals(np.asarray(v, np.float32), np.asarray(f, np.int64)).astype(np.float64)[..., :3]
    t, region, q, d = closest_feature(p, v, f)
    vv = np.asarray(v, np.float64)
    a, b, c = vv[f[t, 0]], vv[f[t, 1]], vv[f[t, 2]]
    n_face = np.cross(b - a, c - a)
    n = np.where((region == 0)[:, None], n_face, pn[t, np.maximum(region - 1, 0)])
    s = np.einsum("ij,ij->i", np.asarray(p, np.float64) - q, n)
    return np.sign(s).astype(int), d
