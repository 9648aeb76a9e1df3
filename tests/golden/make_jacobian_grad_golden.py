"""Golden vectors for the Jacobian-gradient (dJ/dq) branch of the FK backward pass, produced by THE REFERENCE'S OWN CUDA
KERNEL (kinematics_backward_kernel<..., COMPUTE_JACOBIAN_GRAD = true>) run on the CPU through oracle/_ref:

    python tests/golden/make_jacobian_grad_golden.py        (from the repository root, after __graft_entry__.build())

franka and unitree_g1 (four tool frames): q [3, 2, D], weights w on the Jacobian output [3, 2, T, 6, D], and
grad_q = d/dq <w, J(q)>.  Output: tests/golden/jacobian_grad_golden.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conftest import load_model, sample_q  # noqa: E402

from oracle import ref_kernels  # noqa: E402


def main():
    ref = ref_kernels.ReferenceKernels()
    rng = np.random.default_rng(3)
    out = {}
    for robot in ("franka", "unitree_g1"):
        model = load_model(robot)
        md = model.as_dict()
        q = sample_q(model, 6, seed=9, scale=0.7).astype(np.float32)
        fk = ref.kinematics_forward(q, md, compute_jacobian=True)
        T, D = fk["link_pos"].shape[1], q.shape[1]
        w = rng.normal(size=(6, T, 6, D)).astype(np.float32)
        g = ref.kinematics_backward_jacobian(md, fk["cumul_mat"], w)
        out.update({f"{robot}/q": q.reshape(3, 2, D), f"{robot}/w": w.reshape(3, 2, T, 6, D), f"{robot}/grad_q": g.reshape(3, 2, D),
                    f"{robot}/jacobian": fk["jacobian"].reshape(3, 2, T, 6, D)})
        print(robot, "max |grad_q|", float(np.abs(g).max()))
    path = os.path.join(HERE, "jacobian_grad_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
