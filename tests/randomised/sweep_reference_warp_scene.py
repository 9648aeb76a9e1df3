"""The ORACLE's scene collision against the reference's own Warp kernels (run thread by thread through tests/golden/warp_emulator,
as tests/golden/make_scene_warp_golden.py does) on RANDOM worlds: rotated / disabled cuboids, fp16 ESDF grids, two environments,
static and swept, speed metric, negative radii, resting and far-away spheres.  CPU only, needs /root/reference.
    PYTHONPATH=/root/reference python tests/randomised/sweep_reference_warp_scene.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/geom/collision"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, "/root/reference")  # (ahead of the repository root: `curobo` must be the reference's package here, not this repository's facade)
import make_scene_warp_golden as G  # noqa: E402  (the reference's kernels over the Warp stand-in)

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rq():
    q = rng.normal(size=4)
    if rng.random() < 0.25:
        q = np.array([1.0, 0, 0, 0])
    return [float(v) for v in q / np.linalg.norm(q)]


def random_cuboids(n):
    out = []
    for _ in range(n):
        o = {"dims": [float(v) for v in rng.uniform(0.05, 0.6, size=3)], "pose": [float(v) for v in rng.uniform([0.1, -0.4, 0.2], [0.8, 0.4, 0.8])] + rq()}
        if rng.random() < 0.2:
            o["enable"] = False
        out.append(o)
    return out


bad = 0
for case in range(n_cases):
    E = int(rng.integers(1, 3))
    max_n = int(rng.integers(1, 5))
    scene = {}
    kinds = int(rng.integers(1, 4))  # 1 cuboids, 2 voxels, 3 both
    if kinds & 1:
        scene.update(G.cuboid_arrays([random_cuboids(int(rng.integers(1, max_n + 1))) for _ in range(E)], max_n=max_n))
    if kinds & 2:
        grids = [[{"pose": [float(v) for v in rng.uniform([0.3, -0.2, 0.3], [0.6, 0.2, 0.7])] + rq(),
                   "sdf": G.union(G.box_sdf([float(v) for v in rng.uniform(0.03, 0.15, size=3)]), G.ball_sdf(float(rng.uniform(0.04, 0.1)), rng.uniform(-0.1, 0.1, size=3))),
                   "enable": bool(rng.random() < 0.85)}] for _ in range(E)]
        scene.update(G.voxel_arrays(grids, max_n=1, shape=(int(rng.integers(8, 20)), int(rng.integers(8, 20)), int(rng.integers(8, 20))),
                                    voxel_size=float(rng.choice([0.02, 0.03])), max_dist=float(rng.choice([1000.0, 0.5]))))
    B, H, S = int(rng.integers(1, 4)), int(rng.integers(2, 6)), int(rng.integers(1, 8))
    radii = np.array([0.02, 0.035, 0.05, 0.07, -1.0], np.float32)
    sp = G.trajectories(rng, B, H, S, np.array([0.45, 0.0, 0.5]), 0.3, float(rng.choice([0.0, 0.02, 0.06])), radii)
    if rng.random() < 0.5:
        sp[0, :, 0, :3] = sp[0, 0:1, 0, :3]  # a stationary sphere
    env = rng.integers(0, E, size=B).astype(np.int32)
    multi = bool(E > 1 or rng.random() < 0.5)
    if not multi:
        env[:] = 0
    swept = bool(rng.random() < 0.6)
    dt = float(rng.choice([0.02, 0.05])) if rng.random() < 0.5 else None
    w, eta = float(rng.choice([1.0, 2.5, 100.0])), float(rng.choice([0.0, 0.01, 0.03]))
    try:
        want_d, want_g = G.run(sp, scene, w, eta, env, multi, swept, dt)
        r = oracle.scene_collision(sp, scene, w, eta, env, multi, sweep=swept, enable_speed_metric=dt is not None, speed_dt=dt if dt is not None else 0.02)
        scale_d, scale_g = max(1.0, float(want_d.max())), max(1.0, float(np.abs(want_g).max()))
        assert np.array_equal(r["distance"] > 0, want_d > 0), "different spheres in collision"
        np.testing.assert_allclose(r["distance"], want_d, rtol=0, atol=2e-6 * scale_d)
        np.testing.assert_allclose(r["gradient"][..., :3], want_g[..., :3], rtol=0, atol=2e-6 * scale_g)
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: kinds {kinds} E {E} max_n {max_n} B {B} H {H} S {S} swept {swept} dt {dt} w {w} eta {eta}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
