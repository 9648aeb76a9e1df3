"""Golden vectors for the Levenberg-Marquardt step, produced by the REFERENCE's own Warp tile kernel executed on the CPU
through the Warp stand-in of tests/golden/warp_emulator (tile_load / tile_matmul / tile_cholesky / tile_cholesky_solve /
tile_map / tile_sum restated there in fp32, every product and sum rounded in index order):

    PYTHONPATH=/root/reference python tests/golden/make_lm_warp_golden.py

    curobo/_src/optim/util/levenberg_marquardt_step.py   LevenbergMarquardtStep.create_lm_warp_kernel(dof, n_res),
                                                         launched as its forward() does (launch_tiled, dim = problems)

What the reference's source fixes here: delta = -(J^T J + lambda I)^-1 (J^T e), q_out = q_in + delta,
pred_reduction = 0.5 * delta . (lambda delta - J^T e).  Shapes: the seed-IK system (13 residuals x 7 dof) and a
20 x 6 one.  Output: tests/golden/lm_warp_golden.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_scene_warp_golden as _emu  # noqa: E402,F401  (puts the emulator + module stubs in place)

import warp as wp  # noqa: E402

from curobo._src.optim.util.levenberg_marquardt_step import LevenbergMarquardtStep  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    out = {}
    for tag, (n_res, dof, nprob) in {"ik13x7": (13, 7, 48), "r20x6": (20, 6, 32)}.items():
        J = rng.standard_normal((nprob, n_res, dof)).astype(np.float32)
        J[: nprob // 4] *= 0.05  # small Jacobians: the damping term dominates
        e = rng.standard_normal((nprob, n_res)).astype(np.float32)
        jte = np.einsum("prd,pr->pd", J, e).astype(np.float32)
        lam = (10.0 ** rng.uniform(-3, 1, nprob)).astype(np.float32)
        q_in = rng.uniform(-2, 2, (nprob, dof)).astype(np.float32)
        q_out, pred = np.zeros((nprob, dof), np.float32), np.zeros(nprob, np.float32)
        kern = LevenbergMarquardtStep.create_lm_warp_kernel(dof, n_res)
        wp.launch_tiled(kern, dim=[nprob], inputs=[wp.array(J), wp.array(jte), wp.array(lam), wp.array(q_in), wp.array(q_out), wp.array(pred)],
                        block_dim=32)
        # sanity against a float64 solve (not the pin: the pin is the reference's arithmetic above)
        A = np.einsum("prd,pre->pde", J.astype(np.float64), J.astype(np.float64)) + lam[:, None, None] * np.eye(dof)
        d64 = -np.linalg.solve(A, jte.astype(np.float64)[..., None])[..., 0]
        print(tag, "max |delta - float64 solve|", float(np.abs((q_out - q_in) - d64).max()), "pred range", float(pred.min()), float(pred.max()))
        out.update({f"{tag}/jacobian": J, f"{tag}/jTerror": jte, f"{tag}/lambda": lam, f"{tag}/joint_position_in": q_in,
                    f"{tag}/joint_position_out": q_out, f"{tag}/pred_reduction": pred})
    path = os.path.join(HERE, "lm_warp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
