"""Golden vectors produced by THE REFERENCE'S OWN CUDA KERNELS run on the CPU (oracle/_ref/libcurobo_ref.so, built from
/root/reference by `make -C oracle/cuda_on_cpu`, see oracle/ref_kernels.py):

    python tests/golden/make_cuda_kernels_golden.py        (from the repository root, after __graft_entry__.build())

Franka: FK (poses, spheres, cumulative transforms, Jacobian, centre of mass), FK VJP, self collision; B-spline degree 3
with the implicit goal.  Inputs come from the same seeded helpers the tests use, so only the outputs (and the random
gradients) are stored.  Output: tests/golden/cuda_kernels_golden.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/ (conftest helpers)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repository root (oracle package)
from conftest import load_model, sample_q  # noqa: E402
from test_reference_cuda_kernels import KEYS, _bspline_case  # noqa: E402

from oracle import ref_kernels  # noqa: E402


def main():
    ref = ref_kernels.ReferenceKernels()
    model = load_model("franka")
    md = model.as_dict()
    rng = np.random.default_rng(17)
    q = sample_q(model, 16, seed=21)
    fk = ref.kinematics_forward(q, md, compute_jacobian=True, compute_com=True)
    out = {"fk/q": q, **{"fk/" + k: v for k, v in fk.items()}}
    n, S, T = q.shape[0], fk["robot_spheres"].shape[1], fk["link_pos"].shape[1]
    gs = rng.standard_normal((n, S, 4)).astype(np.float32)
    gs[..., 3] = 0
    gp, gq = rng.standard_normal((n, T, 3)).astype(np.float32), rng.standard_normal((n, T, 4)).astype(np.float32)
    out.update({"bwd/grad_spheres": gs, "bwd/grad_link_pos": gp, "bwd/grad_link_quat": gq,
                "bwd/grad_q": ref.kinematics_backward(md, fk["cumul_mat"], gs, gp, gq)})
    sph = ref.kinematics_forward(sample_q(model, 24, seed=13, scale=1.3), md)["robot_spheres"]
    sc = ref.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.5)
    out.update({"self/spheres": sph, **{"self/" + k: v for k, v in sc.items()}})
    fwd, bwd = _bspline_case(3, 1)
    bs = ref.bspline_forward(*fwd)
    out.update({"bspline/" + k: bs[k] for k in KEYS})
    out["bspline/grad_knots"] = ref.bspline_backward(*bwd)
    path = os.path.join(HERE, "cuda_kernels_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "self-collision hits", int((sc["distance"] > 0).sum()))


if __name__ == "__main__":
    main()
