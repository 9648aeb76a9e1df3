"""Golden vectors for the RNEA oracle, produced by the REFERENCE's own NumPy implementation
(curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py: the in-tree oracle the reference
validates its CUDA kernel against, test_rnea_reference.py).  Run in the build container
(needs /root/reference):   python tests/golden/make_rnea_golden.py
Writes tests/golden/rnea_golden.npz: inputs + tau, v, a, f and the VJP for franka and unitree_g1.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/curobo/tests/_src/robot/dynamics/rnea_numpy_reference.py"


def main():
    spec = importlib.util.spec_from_file_location("rnea_numpy_reference", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from curobo_amd.robot import load_packaged_robot

    out = {}
    for seed, (robot, n) in enumerate((("franka", 6), ("unitree_g1", 3))):
        m = load_packaged_robot(robot).as_dict()
        L = m["fixed_transforms"].shape[0]
        dof = int(m["num_dof"])
        rng = np.random.default_rng(100 + seed)
        lo, hi = m["joint_limits_position"]
        q = rng.uniform(lo, hi, size=(n, dof))
        qd = rng.normal(size=(n, dof))
        qdd = rng.normal(size=(n, dof)) * 2.0
        tau_bar = rng.normal(size=(n, dof))
        off = np.asarray(m["joint_offset_map"], np.float64).reshape(L, 2)
        args = dict(fixed_transforms=np.asarray(m["fixed_transforms"], np.float64).reshape(L, 3, 4),
                    link_map=m["link_map"], joint_map=m["joint_map"], joint_map_type=m["joint_map_type"],
                    joint_offset_map=off, link_masses_com=np.asarray(m["link_masses_com"], np.float64),
                    link_inertias=np.asarray(m["link_inertias"], np.float64)[:, :6])
        res = {k: [] for k in ("tau", "v", "a", "f", "grad_q", "grad_qd", "grad_qdd")}
        for i in range(n):
            tau, v, a, f = ref.rnea(q[i], qd[i], qdd[i], gravity=-9.81, **args)
            gq, gqd, gqdd = ref.rnea_backward(tau_bar[i], q[i], qd[i], qdd[i], v, a, f, gravity=-9.81, **args)
            for k, val in zip(res, (tau, v, a, f, gq, gqd, gqdd)):
                res[k].append(val)
        for k, val in (("q", q), ("qd", qd), ("qdd", qdd), ("tau_bar", tau_bar)):
            out[f"{robot}/{k}"] = val
        for k, val in res.items():
            out[f"{robot}/{k}"] = np.stack(val)
    np.savez_compressed(os.path.join(HERE, "rnea_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "rnea_golden.npz"), {k: v.shape for k, v in out.items() if k.startswith("franka")})


if __name__ == "__main__":
    main()
