"""The mesh launch (select + group walk) against the oracle's brute force on random mesh worlds: boxes (thin slabs and pillars
among them), spheres, tori, L prisms at random poses and sizes, disabled slots, discrete / swept / speed metric, random
activation distance.   python tests/randomised/fuzz_mesh.py [cases] [seed] [--open] [--deep]

--deep (end of round 6): a third of the spheres of every case are moved to within centimetres of a mesh's middle -- the points
about equally far from very many triangles, which the cell-list kernel hands to a workgroup each (sphere_mesh_wide_kernel).

--open (round 6): a third of the meshes lose faces or get some flipped; the oracle signs EVERY mesh with the reference's rule
(Warp's three axis rays, ``set_mesh_sign_rule("rays")``: on the closed meshes of the world that is the same function), the device
picks its rule per mesh from the topology.  A sphere whose ray grazes an edge may differ (fp32 rays against fp64): a small
allowance of whole spheres."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402
from test_oracle_mesh import box_shape, ell_shape, sphere_shape, torus_shape  # noqa: E402

from curobo_amd.backends import collision as Cn  # noqa: E402
from curobo_amd.scene import SceneData  # noqa: E402
from oracle.oracle import Oracle, mesh_scene_arrays  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
OPEN, DEEP = "--open" in sys.argv, "--deep" in sys.argv
sys.argv = [a for a in sys.argv if a not in ("--open", "--deep")]
n_wide = n_walk = 0
if OPEN:
    oracle.set_mesh_sign_rule("rays")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
model = load_model("franka")


def rq():
    q = rng.normal(size=4)
    return [float(v) for v in q / np.linalg.norm(q)]


def random_mesh(i):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        v, f = box_shape([float(x) for x in rng.uniform(0.05, 0.6, size=3)], int(rng.integers(0, 4)))
    elif kind == 1:  # a thin pillar / slab: the long walks of the bench's mesh world
        d = [0.1, 0.1, 0.1]
        d[int(rng.integers(3))] = float(rng.uniform(0.8, 1.6))
        v, f = box_shape(d, int(rng.integers(2, 5)))
    elif kind == 2:
        v, f = sphere_shape(float(rng.uniform(0.05, 0.3)), int(rng.choice([6, 12, 24])), int(rng.choice([8, 24, 48])))
    elif kind == 3:
        v, f = torus_shape(float(rng.uniform(0.15, 0.3)), float(rng.uniform(0.03, 0.08)), int(rng.choice([16, 48])), int(rng.choice([8, 24])))
    else:
        v, f = ell_shape(int(rng.integers(1, 3)))
    if OPEN and rng.random() < 0.35:
        f = np.asarray(f).copy()
        if rng.random() < 0.5:  # an open mesh: a few faces missing
            f = f[rng.random(len(f)) > float(rng.uniform(0.02, 0.3))]
        else:                   # inconsistently oriented: a few faces wound the other way
            flip = rng.random(len(f)) < float(rng.uniform(0.01, 0.2))
            f[flip] = f[flip][:, [0, 2, 1]]
    o = {"name": f"m{i}", "vertices": v, "faces": f, "pose": [float(x) for x in rng.uniform([-0.6, -0.6, -0.1], [0.6, 0.6, 0.9])] + rq()}
    if rng.random() < 0.15:
        o["enable"] = False
    return o


bad = 0
for case in range(n_cases):
    sweep = bool(rng.random() < 0.5)
    speed = sweep and bool(rng.random() < 0.6)
    eta = float(rng.choice([0.0, 0.0025, 0.02, 0.08]))
    world = [[random_mesh(i) for i in range(int(rng.integers(1, 7)))]]
    b, h = int(rng.integers(1, 24)), int(rng.integers(2, 12))
    q0, q1 = sample_q(model, b, seed=int(rng.integers(1000)))[:, None], sample_q(model, b, seed=int(rng.integers(1000)))[:, None]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    sph = oracle.kinematics_forward((q0 * (1 - tt) + q1 * tt).reshape(b * h, -1) * float(rng.uniform(0.3, 1.0)), model.as_dict(), horizon=h)["robot_spheres"]
    sph = sph.reshape(b, h, -1, 4)
    S = sph.shape[2]
    pick = np.zeros((b, h, S), bool)
    if DEEP:
        sph = sph.copy()
        centres = np.asarray([m["pose"][:3] for m in world[0]], np.float32)
        pick = rng.random((b, h, S)) < 0.33
        sph[pick, :3] = centres[rng.integers(0, len(centres), size=int(pick.sum()))] + rng.normal(size=(int(pick.sum()), 3)).astype(np.float32) * float(rng.choice([0.002, 0.02, 0.06]))
    try:
        ref = oracle.scene_collision(sph, mesh_scene_arrays(world), 3.0, eta, sweep=sweep, enable_speed_metric=speed, speed_dt=0.05)
        scene = SceneData.from_arrays(None, dev, meshes=world)
        dist, grad = torch.full((b, h, S), 5.0, device=dev), torch.full((b, h, S, 4), 5.0, device=dev)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=dev), scene.struct, torch.tensor([3.0], device=dev),
                                     torch.tensor([eta], device=dev), None, b, h, S, False, 3 if sweep else 0, speed, torch.tensor([0.05], device=dev))
        torch.cuda.synchronize()
        cnt = next(iter(dist._curobo_mesh_ws.values()))[:16].view(torch.int32).tolist() if hasattr(dist, "_curobo_mesh_ws") else [0] * 4
        n_walk, n_wide = n_walk + cnt[1], n_wide + cnt[3]
        d, g = dist.cpu().numpy(), grad.cpu().numpy()
        dr, gr = ref["distance"], ref["gradient"]
        ok = np.ones(d.shape, bool)
        if sweep:
            stepn = np.linalg.norm(np.diff(sph[..., :3], axis=1), axis=-1)
            ok[:, 1:] &= stepn >= 1e-5
            ok[:, :-1] &= stepn >= 1e-5
        sc = 20.0 if speed else 1.0
        graze = np.abs(d - dr) < 3e-5 * sc
        if OPEN:
            assert ((d > 0) != (dr > 0))[ok & ~graze].sum() <= 2 + int(4e-3 * (dr > 0).sum()), "hit set differs"
        else:
            assert np.array_equal((d > 0)[ok & ~graze], (dr > 0)[ok & ~graze]), "hit set differs"
        e, tol = np.abs(d - dr)[ok], 3e-5 * sc + 2e-4 * np.abs(dr)[ok]
        n_off = int((e > tol).sum())
        allowed = (2 + int(2e-4 * (dr > 0).sum())) if sweep else 0
        if OPEN:
            allowed += 2 + int(4e-3 * (dr > 0).sum())  # (a ray through an edge: the sign of a whole sphere)
        assert n_off <= allowed, f"{n_off} spheres beyond the cost bound (allowed {allowed}), worst {float((e / tol).max()):.1f} x; colliding {int((dr > 0).sum())}"
        # (--deep: in the middle of a mesh the closest triangle is one of many within rounding of each other: the moved spheres'
        #  distances are held, their gradients are not)
        okg = ok & ~pick
        badg = np.abs(g - gr).max(-1)[okg] > (3e-4 * sc + 2e-3 * np.abs(gr).max(-1)[okg])
        assert badg.size == 0 or badg.mean() < 3e-3, f"gradient: {float(badg.mean()):.2e} of the spheres off (closest-point ties aside)"
    except AssertionError as ex:
        bad += 1
        print(f"FAILED case {case}: meshes {[(m['name'], len(m['faces'])) for m in world[0]]} sweep {sweep} speed {speed} eta {eta} b {b} h {h}: {str(ex)[:300]}")
print(f"{n_cases} cases, {bad} failed  (spheres handed to the tree walk {n_walk}, to a workgroup of their own {n_wide})")
