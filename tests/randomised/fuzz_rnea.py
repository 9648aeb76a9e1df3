"""RNEA forward / VJP against the oracle: robots, batch sizes around the wavefront and workgroup boundaries, velocity and
acceleration scales, external forces, gravity directions, the staged / lane / quad / scratch launch forms.
    python tests/randomised/fuzz_rnea.py [cases] [seed]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--worker" not in sys.argv:  # the launch form is chosen by environment variables read once per process: one worker per form
    n_cases, seed = (sys.argv[1] if len(sys.argv) > 1 else "12"), (sys.argv[2] if len(sys.argv) > 2 else "1")
    total = 0
    for env in ({}, {"CUROBO_RNEA_QUAD": "0"}, {"CUROBO_RNEA_STAGED": "0"}, {"CUROBO_RNEA_SCRATCH_QUAD": "1"}):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), n_cases, seed, "--worker"], env={**os.environ, **env}, capture_output=True, text=True)
        tail = [l for l in out.stdout.splitlines() if l.startswith(("RNEA", "FAILED"))]
        print(env or "default", "->", *tail, sep="\n  ")
        if out.returncode != 0:
            print(out.stderr[-1500:])
            total += 1
        total += sum(l.startswith("FAILED") for l in tail)
    print("failed in total:", total)
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402

from curobo_amd.backends import dynamics as Dy  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_cases, rng = int(sys.argv[1]), np.random.default_rng(int(sys.argv[2]))
models = {r: load_model(r) for r in ("franka", "ur10e", "unitree_g1")}
kins = {r: KinematicsParams.from_model(m, dev) for r, m in models.items()}
t = lambda a: None if a is None else torch.as_tensor(a, device=dev)  # noqa: E731
bad = 0
for case in range(n_cases):
    robot = str(rng.choice(list(models)))
    model, kin = models[robot], kins[robot]
    n = int(rng.choice([1, 3, 63, 64, 65, 255, 256, 257, 1000]))
    md, L, D = model.as_dict(), kin.num_links, kin.num_dof
    q = sample_q(model, n, seed=int(rng.integers(10000))).astype(np.float32) * float(rng.choice([0.3, 1.0, 2.0]))
    qd = (rng.normal(size=q.shape) * float(rng.choice([0.0, 1.0, 10.0]))).astype(np.float32)
    qdd = (rng.normal(size=q.shape) * float(rng.choice([0.0, 2.0, 50.0]))).astype(np.float32)
    with_fe = bool(rng.random() < 0.4)
    fe = rng.normal(size=(n, L, 6)).astype(np.float32) if with_fe else None
    grav = [np.array([0, 0, 0, 0, 0, 9.81], np.float32), np.zeros(6, np.float32), np.array([0, 0, 0, 3.0, -4.0, 5.0], np.float32)][int(rng.integers(3))]
    scratch = torch.zeros(3 * n * D, device=dev) if (rng.random() < 0.4 and not with_fe) else None
    try:
        tau_ref, cache_ref = oracle.rnea_forward(q, qd, qdd, md, gravity=grav, f_ext=fe)
        tau, cache = torch.zeros(n, D, device=dev), torch.zeros(n, L * 20, device=dev)
        args = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
                kin.joint_offset_map, t(grav), kin.link_level_offsets, kin.link_level_data)
        Dy.launch_rnea_forward(tau, t(q), t(qd), t(qdd), *args, cache, n, L, D, kin.n_tree_levels, 1, t(fe), scratch=scratch)
        torch.cuda.synchronize()
        np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(tau_ref).max())))
        w = rng.normal(size=(n, D)).astype(np.float32)
        ref_g = oracle.rnea_backward(w, q, qd, cache_ref, md, gravity=grav, want_f_ext_grad=with_fe)
        g = [torch.full((n, D), 7.0, device=dev) for _ in range(3)]
        gfe = torch.zeros(n, L, 6, device=dev) if with_fe else None
        ws = torch.zeros(n, L * 18, device=dev)
        if scratch is not None:
            Dy.launch_rnea_backward(*g, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, gfe, ws, scratch=scratch, scratch_holds_q_qd=True)
        else:
            Dy.launch_rnea_backward(*g, t(w), t(q), t(qd), *args, cache, n, L, D, kin.n_tree_levels, 1, gfe, ws)
        torch.cuda.synchronize()
        for ours, ref in zip(g, ref_g[:3]):
            np.testing.assert_allclose(ours.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(ref).max())))
        if with_fe:
            np.testing.assert_allclose(gfe.cpu().numpy(), ref_g[3], rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(ref_g[3]).max())))
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: {robot} n {n} f_ext {with_fe} scratch {scratch is not None}: {str(e)[:300]}".replace("\n", " | "))
print(f"RNEA cases: {n_cases}, failed: {bad}")
