"""Helpers of the swept-collision parity tests (test infrastructure).

The reference's swept kernel adds a duplicate of the centre sample of a direction iff the half sweep length in the obstacle
frame is > 0 (wp_sweep_collision_kernel.py:186-203).  A sphere that is stationary up to rounding therefore costs 1x / 2x /
3x its centre cost depending on the LAST BIT of the world -> obstacle-frame transform.  Two ways to hold the HIP path to the
oracle in spite of that, both used by test_gpu_fused.py / test_gpu_parity_benchmarked.py:

* ``device_frame_arithmetic(oracle)``: the oracle transforms with the device's arithmetic (orc_set_frame_arithmetic(1)); fed
  the device's own spheres, its `half_dist > 0` decisions are the device's and EVERY trajectory is compared at 1e-5;
* ``per_sphere_allowance(...)``: the oracle in the reference's arithmetic; every sphere without a stationary neighbour is
  compared tightly, a sphere with n stationary neighbours may differ by k_o whole centre-sample terms of obstacle o,
  |k_o| <= n, and by nothing else.
"""

import contextlib
import itertools

import numpy as np


@contextlib.contextmanager
def device_frame_arithmetic(oracle):
    oracle.set_frame_arithmetic("device")
    try:
        yield oracle
    finally:
        oracle.set_frame_arithmetic("reference")


def still_neighbours(sph, tol=1e-5):
    """[B, H, S] number of neighbour points (0 .. 2) a sphere does not move towards, up to rounding (world frame)"""
    p = sph[..., :3]
    stepn = np.linalg.norm(np.diff(p, axis=1), axis=-1)
    n = np.zeros(p.shape[:3], np.int32)
    n[:, 1:] += stepn < tol
    n[:, :-1] += stepn < tol
    return n


def rest_in_collision(sph, scene_cost):
    """[B] trajectories that hold a sphere which is stationary up to rounding AND in collision"""
    return ((still_neighbours(sph) > 0) & (scene_cost > 0)).any(axis=(1, 2))


def per_obstacle_centre_terms(oracle, sph, arrays, w, eta, env_idx, speed_dt):
    """Centre-sample cost / gradient of every sphere against every obstacle ALONE (sweep off, speed metric on: the map is
    linear in (cost, gradient), so this is the term a duplicated centre sample adds): lists over the obstacles."""
    out = []
    n_c = arrays["cuboid_dims"].shape[1] if arrays.get("cuboid_dims") is not None else 0
    n_v = arrays["voxel_params"].shape[1] if arrays.get("voxel_params") is not None else 0
    multi = env_idx is not None
    for n, key in ((n_c, "cuboid_enable"), (n_v, "voxel_enable")):
        for o in range(n):
            part = dict(arrays)
            for k2 in ("cuboid_enable", "voxel_enable"):
                if part.get(k2) is not None:
                    part[k2] = np.zeros_like(arrays[k2])
            part[key] = np.zeros_like(arrays[key])
            part[key][:, o] = arrays[key][:, o]
            r = oracle.scene_collision(sph, part, w, eta, sweep=False, enable_speed_metric=True, speed_dt=speed_dt,
                                       env_query_idx=env_idx, use_multi_env=multi)
            out.append((r["distance"], r["gradient"][..., :3]))
    return out


def per_sphere_allowance(oracle, d, g, d_ref, g_ref, sph, arrays, w, eta, env_idx, speed_dt, max_split, tag,
                         tol_abs=5e-6, min_colliding=1000):
    """Per-sphere scene cost ``d`` / gradient ``g`` [B, H, S(, 3)] of a HIP kernel against the oracle's (reference
    arithmetic) on the SAME spheres.  Asserts: which moving spheres collide is identical; every sphere without a stationary
    neighbour within 1e-5 relative + ``tol_abs`` m of penetration, except at most ``max_split`` whose sweep took one sample
    more or less (`if jump >= half_dist: break` within rounding of the half segment); ambiguous spheres differ by whole
    centre-sample terms only.  Returns {corr, split, amb, frac_amb, flipped}: ``corr`` is what the HIP branches add to the
    oracle's per-sphere cost."""
    n_still = still_neighbours(sph)
    amb = (n_still > 0) & ((d > 0) | (d_ref > 0))
    frac_amb = float(amb.mean())
    in_col = d_ref > 0
    assert in_col.sum() > min_colliding, "the workload must collide"
    assert np.array_equal((d > 0) & ~amb, in_col & ~amb), f"[{tag}] which moving spheres collide must be identical"
    tight = ~amb
    e_d = np.abs(d[tight] - d_ref[tight])
    tol_d = 1e-5 * np.abs(d_ref[tight]) + tol_abs * w
    e_g = np.abs(g[tight] - g_ref[tight])
    tol_g = 1e-3 * np.abs(g_ref[tight]) + 2e-4 * w
    print(f"\n[{tag} per sphere] {d.shape[0]} trajectories, {int(in_col.sum())} colliding spheres, ambiguous (stationary and in "
          f"collision) {int(amb.sum())} = {frac_amb:.2e} of all spheres, in {int(amb.any((1, 2)).sum())} trajectories; tight spheres: "
          f"max cost error {float((e_d / tol_d).max()):.3f} of the bound ({float(e_d.max() / w):.2e} m), gradient "
          f"{float((e_g / tol_g).max()):.3f} of the bound")
    split = np.zeros(d.shape, bool)
    split[tight] = e_d > tol_d
    n_split = int(split.sum())
    assert n_split <= max_split, f"[{tag}] tight spheres: {n_split} beyond the bound (allowed {max_split}); ambiguous fraction {frac_amb:.2e}"
    if n_split:
        print(f"[{tag} per sphere] moving spheres whose sweep took one sample more / less than the oracle's: {n_split} of "
              f"{int(in_col.sum())} colliding (allowed {max_split}); largest difference {float(e_d.max() / w):.2e} m")
    ok_g = ~(split[tight])
    assert (e_g[ok_g] > tol_g[ok_g]).mean() < 1e-5 and (e_g[ok_g] <= 30 * tol_g[ok_g]).all(), \
        (tag, int((e_g[ok_g] > tol_g[ok_g]).sum()), float((e_g[ok_g] / tol_g[ok_g]).max()), f"ambiguous fraction {frac_amb:.2e}")
    corr = np.zeros_like(d_ref, dtype=np.float64)
    flipped = 0
    if amb.any():
        terms = per_obstacle_centre_terms(oracle, sph, arrays, w, eta, env_idx, speed_dt)
        ia = np.nonzero(amb)
        diff = (d[ia] - d_ref[ia]).astype(np.float64)
        c1 = np.stack([t[0][ia] for t in terms], axis=1).astype(np.float64)  # [n_amb, n_obs]
        nmax = n_still[ia]
        best = np.full(diff.shape, np.inf)
        best_k = np.zeros_like(c1)
        # (fewest whole terms first: an obstacle the sphere does not touch has c1 = 0 and must not collect a k)
        for ks in sorted(itertools.product(range(-2, 3), repeat=c1.shape[1]), key=lambda k: sum(abs(v) for v in k)):
            kv = np.asarray(ks, np.float64)
            ok = (np.abs(kv)[None, :] <= nmax[:, None]).all(1)
            r = np.where(ok, np.abs(diff - c1 @ kv), np.inf)
            better = r < best - 1e-12
            best = np.where(better, r, best)
            best_k[better] = kv
        tol_a = 1e-5 * (np.abs(d_ref[ia]) + np.abs(c1).sum(1)) + tol_abs * w
        assert (best <= tol_a).all(), (f"[{tag}] {int((best > tol_a).sum())} of {amb.sum()} ambiguous spheres differ from the oracle by more "
                                      f"than whole centre-sample terms (ambiguous fraction {frac_amb:.2e})")
        corr[ia] = (c1 * best_k).sum(1)
        flipped = int((np.abs(c1 * best_k).sum(1) > 0).sum())
        print(f"[{tag} per sphere] ambiguous spheres: {int(amb.sum())}, of which the HIP kernel and the oracle took different sweep "
              f"branches (a non-zero whole-term correction): {flipped}; largest residual {float((best / tol_a).max()):.3f} of the bound")
    return {"corr": corr, "split": split, "amb": amb, "frac_amb": frac_amb, "flipped": flipped, "n_still": n_still}


def assert_scene_kernel_parity(oracle, dist, grad, sph, arrays, w, eta, tag, voxel=False, **kw):
    """A scene-collision KERNEL's per-sphere cost ``dist`` [B, H, S] / gradient ``grad`` [B, H, S, 4] against the oracle on the
    same spheres at north_star's tolerance: hit set EXACT (bit-exact collision indices), cost 1e-5 relative + 1e-6 m of
    penetration for closed-form obstacles (5e-6 m with an fp16 ESDF: up to eight trilinear samples of half-precision data
    per lookup, up to seven lookups per swept sphere), gradient 1e-3 relative + 2e-4 of the weight (north_star prices poses
    and costs only; the gradient of the quadratic zone divides a 1e-7 m rounding by eta).  The oracle's obstacle-frame
    transform runs in the device's arithmetic, so sweep decisions at zero motion cannot split the two; the same comparison
    in the reference's arithmetic is printed next to it.  ``kw`` -> oracle.scene_collision (sweep, enable_speed_metric ...)."""
    with device_frame_arithmetic(oracle):
        ref = oracle.scene_collision(sph, arrays, w, eta, **kw)
    ref0 = oracle.scene_collision(sph, arrays, w, eta, **kw)
    d = np.asarray(dist)
    g = np.asarray(grad)
    tol_abs = (5e-6 if voxel else 1e-6) * w

    def worst(r):
        return float((np.abs(d - r["distance"]) / (1e-5 * np.abs(r["distance"]) + tol_abs)).max())

    gerr = float((np.abs(g - ref["gradient"]) / (1e-3 * np.abs(ref["gradient"]) + 2e-4 * w)).max())
    print(f"\n[{tag}] {int((ref['distance'] > 0).sum())} colliding spheres of {d.size}; worst cost error {worst(ref):.3f} of the bound "
          f"(1e-5 rel + {tol_abs / w:.0e} m) in the device's frame arithmetic, {worst(ref0):.3f} in the reference's; gradient {gerr:.3f} of its bound")
    assert np.array_equal(d > 0, ref["distance"] > 0), f"[{tag}] which spheres collide must be identical"
    np.testing.assert_allclose(d, ref["distance"], rtol=1e-5, atol=tol_abs)
    np.testing.assert_allclose(g, ref["gradient"], rtol=1e-3, atol=2e-4 * w)
    return ref
