"""FK forward / VJP and the B-spline kernels against the oracle on random batch sizes, horizons, joint ranges (inside, at and
beyond the limits, many turns) and spline shapes.   python tests/randomised/fuzz_fk_bspline.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402

from curobo_amd.backends import kinematics as K  # noqa: E402
from curobo_amd.backends import trajectory as Tr  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
bad = 0
models = {r: load_model(r) for r in ("franka", "ur10e", "unitree_g1")}
kps = {r: KinematicsParams.from_model(m, dev) for r, m in models.items()}
for case in range(n_cases):
    robot = str(rng.choice(list(models)))
    model, kp = models[robot], kps[robot]
    n = int(rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 255, 257, 1000]))
    h = int(rng.choice([1, 1, 3, 33])) if n % 3 == 0 and n > 2 else 1
    if n % h:
        h = 1
    scale = float(rng.choice([0.2, 1.0, 1.0, 3.0, 40.0]))
    q = (sample_q(model, n, seed=int(rng.integers(10000))) * scale).astype(np.float32)
    S, T, L, d = model.num_spheres, len(model.tool_frames), model.num_links, model.num_dof
    try:
        ref = oracle.kinematics_forward(q, model.as_dict(), compute_com=True, horizon=h)
        o = dict(link_pos=torch.zeros(n, T, 3, device=dev), link_quat=torch.zeros(n, T, 4, device=dev), spheres=torch.zeros(n, S, 4, device=dev),
                 com=torch.zeros(n, 4, device=dev), cumul=torch.zeros(n, L, 3, 4, device=dev))
        env = torch.zeros(n // h, dtype=torch.int32, device=dev)
        K.launch_kinematics_forward_spheres(o["link_pos"], o["link_quat"], o["spheres"], o["com"], o["cumul"], t(q), kp.fixed_transforms,
                                            kp.link_spheres, kp.link_masses_com, kp.joint_map_type, kp.joint_map, kp.link_map, kp.tool_frame_map,
                                            kp.link_sphere_idx_map, kp.joint_offset_map, env, kp.num_envs, n, h, d, S, 32, True, True)  # (batch_size = points)
        torch.cuda.synchronize()
        tol = 1e-5 * max(1.0, scale / 3.0)  # (the angle itself carries 2^-24 of its magnitude)
        np.testing.assert_allclose(o["link_pos"].cpu().numpy().reshape(n, T, 3), ref["link_pos"].reshape(n, T, 3), atol=tol, rtol=0)
        np.testing.assert_allclose(o["spheres"].cpu().numpy().reshape(n, S, 4), ref["robot_spheres"].reshape(n, S, 4), atol=tol, rtol=0)
        np.testing.assert_allclose(o["cumul"].cpu().numpy().reshape(n, L, 3, 4), ref["cumul_mat"].reshape(n, L, 3, 4), atol=tol, rtol=0)
        # quaternions: the same rotation (sign)
        qa, qb = o["link_quat"].cpu().numpy().reshape(n, T, 4), ref["link_quat"].reshape(n, T, 4)
        assert (np.abs(np.abs((qa * qb).sum(-1)) - 1.0) < 1e-5 * max(1.0, scale)).all(), "tool-frame quaternion"
        # VJP on the oracle's transforms
        g_s = rng.normal(size=(n, S, 4)).astype(np.float32)
        g_s[rng.uniform(size=(n, S)) < float(rng.choice([0.0, 0.6, 0.99]))] = 0.0
        g_p, g_q, g_c = rng.normal(size=(n, T, 3)).astype(np.float32), rng.normal(size=(n, T, 4)).astype(np.float32), rng.normal(size=(n, 4)).astype(np.float32)
        cm = ref["cumul_mat"].reshape(n, L, 3, 4)
        com = ref["com"].reshape(n, 4)
        refb = oracle.kinematics_backward(model.as_dict(), cm, g_s, g_p, g_q, g_c, com)
        out = torch.zeros(n, d, device=dev)
        K.launch_kinematics_backward(out, t(g_p), t(g_q), t(g_s), t(g_c), t(com), t(g_p), t(cm), kp.link_spheres, kp.link_masses_com, kp.link_map,
                                     kp.joint_map, kp.joint_map_type, kp.tool_frame_map, kp.link_sphere_idx_map, kp.link_chain_data,
                                     kp.link_chain_offsets, kp.joint_links_data, kp.joint_links_offsets, kp.joint_affects_endeffector,
                                     kp.joint_offset_map, torch.zeros(n, dtype=torch.int32, device=dev), kp.num_envs, n, 1, d, S, True, False)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), refb, atol=2e-5 * max(np.abs(refb).max(), 1.0), rtol=2e-4)
    except AssertionError as e:
        bad += 1
        print(f"FK FAILED case {case}: {robot} n {n} h {h} scale {scale}: {str(e)[:400]}".replace("\n", " | "))
print("FK cases:", n_cases, "failed:", bad)
# ---------------------------------------------------------------- B-spline forward / VJP
keys = ("position", "velocity", "acceleration", "jerk")
for case in range(n_cases):
    degree = int(rng.choice([3, 4, 5]))
    b, nk, dof, interp = int(rng.choice([1, 2, 19, 64, 257])), int(rng.choice([1, 2, 4, 12, 30])), int(rng.choice([1, 6, 7, 49])), int(rng.choice([1, 2, 5]))
    ph = (nk + degree + 1) * interp + 1
    implicit = bool(rng.random() < 0.5)
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    ns, ng = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    mk = lambda n: {k: rng.normal(size=(n, dof)).astype(np.float32) * 0.3 for k in keys}  # noqa: E731
    start, goal = mk(ns), mk(ng)
    sidx, gidx = rng.integers(0, ns, size=b).astype(np.int32), rng.integers(0, ng, size=b).astype(np.int32)
    dt = rng.uniform(0.005, 0.3, size=ng).astype(np.float32)
    imp = np.full(ng, implicit, np.uint8)
    try:
        ref = oracle.bspline_forward(u, start, goal, sidx, gidx, dt, imp, ph, degree)
        outs = [torch.zeros(b, ph, dof, device=dev) for _ in range(4)]
        out_dt = torch.zeros(b, device=dev)
        Tr.launch_bspline_interpolation_forward_kernel(*outs, out_dt, t(u), *[t(start[k]) for k in keys], *[t(goal[k]) for k in keys], t(sidx), t(gidx),
                                                       t(dt), t(imp), b, ph, dof, nk, degree)
        torch.cuda.synchronize()
        for o_, k in zip(outs, keys):
            sc = max(1.0, float(np.abs(ref[k]).max()))
            np.testing.assert_allclose(o_.cpu().numpy(), ref[k], atol=2e-5 * sc, rtol=2e-5, err_msg=k)
        g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
        refb = oracle.bspline_backward(*g, dt, gidx, imp, nk, degree)
        og = torch.zeros(b, nk, dof, device=dev)
        Tr.launch_bspline_interpolation_backward_kernel(og, *[t(x) for x in g], t(dt), t(gidx), t(imp), b, ph, dof, nk, degree, False)
        torch.cuda.synchronize()
        np.testing.assert_allclose(og.cpu().numpy(), refb, atol=2e-5 * max(1.0, float(np.abs(refb).max())), rtol=2e-4)
    except (AssertionError, ValueError) as e:
        bad += 1
        print(f"B-spline FAILED case {case}: degree {degree} b {b} knots {nk} dof {dof} interp {interp} implicit {implicit}: {type(e).__name__} {str(e)[:300]}".replace("\n", " | "))
print("B-spline cases:", n_cases, "failed in total:", bad)
