"""The oracle's RNEA forward / VJP against the reference's own NumPy implementation
(curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py, imported as tests/golden/make_rnea_golden.py does) on random
configurations, velocity / acceleration scales and robots.  CPU only.
    python tests/randomised/sweep_reference_numpy_rnea.py [cases] [seed]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py"
if not os.path.isfile(REF):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("rnea_numpy_reference", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
from conftest import load_model  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
models = {r: load_model(r).as_dict() for r in ("franka", "ur10e", "unitree_g1")}
bad = 0
for case in range(n_cases):
    robot = str(rng.choice(list(models), p=[0.45, 0.35, 0.2]))
    m = models[robot]
    L, dof = m["fixed_transforms"].shape[0], int(m["num_dof"])
    n = int(rng.integers(1, 5 if robot != "unitree_g1" else 3))
    lo, hi = m["joint_limits_position"]
    q = rng.uniform(lo, hi, size=(n, dof)) * float(rng.choice([0.3, 1.0, 1.5]))
    qd = rng.normal(size=(n, dof)) * float(rng.choice([0.0, 1.0, 6.0]))
    qdd = rng.normal(size=(n, dof)) * float(rng.choice([0.0, 2.0, 20.0]))
    tau_bar = rng.normal(size=(n, dof))
    off = np.asarray(m["joint_offset_map"], np.float64).reshape(L, 2)
    args = dict(fixed_transforms=np.asarray(m["fixed_transforms"], np.float64).reshape(L, 3, 4), link_map=m["link_map"], joint_map=m["joint_map"],
                joint_map_type=m["joint_map_type"], joint_offset_map=off, link_masses_com=np.asarray(m["link_masses_com"], np.float64),
                link_inertias=np.asarray(m["link_inertias"], np.float64)[:, :6])
    try:
        want = {k: [] for k in ("tau", "v", "a", "f", "gq", "gqd", "gqdd")}
        for i in range(n):
            tau, v, a, f = ref.rnea(q[i], qd[i], qdd[i], gravity=-9.81, **args)
            g3 = ref.rnea_backward(tau_bar[i], q[i], qd[i], qdd[i], v, a, f, gravity=-9.81, **args)
            for k, val in zip(want, (tau, v, a, f, *g3)):
                want[k].append(val)
        want = {k: np.stack(v) for k, v in want.items()}
        qf, qdf, qddf = q.astype(np.float32), qd.astype(np.float32), qdd.astype(np.float32)
        tau, cache = oracle.rnea_forward(qf, qdf, qddf, m)
        np.testing.assert_allclose(tau, want["tau"], rtol=3e-4, atol=3e-5 * max(1.0, float(np.abs(want["tau"]).max())), err_msg="tau")
        for name, sl in (("v", slice(0, 6)), ("a", slice(6, 12)), ("f", slice(12, 18))):
            np.testing.assert_allclose(cache[:, :, sl], want[name], rtol=3e-4, atol=3e-5 * max(1.0, float(np.abs(want[name]).max())), err_msg=name)
        got = oracle.rnea_backward(tau_bar.astype(np.float32), qf, qdf, cache, m)
        for ours, name in zip(got, ("gq", "gqd", "gqdd")):
            np.testing.assert_allclose(ours, want[name], rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(want[name]).max())), err_msg=name)
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: {robot} n {n}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
