"""The ORACLE's Levenberg-Marquardt step against the reference's own Warp tile kernel (LevenbergMarquardtStep.create_lm_warp_kernel
through tests/golden/warp_emulator) on random shapes: 1-12 dof, residual counts 1-30, damping 1e-6 .. 1e3, tiny / large Jacobians,
rank-deficient ones.  CPU only, needs /root/reference.   python tests/randomised/sweep_reference_warp_lm.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/optim"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, "/root/reference")
import make_scene_warp_golden as _emu  # noqa: E402,F401
import warp as wp  # noqa: E402
from curobo._src.optim.util.levenberg_marquardt_step import LevenbergMarquardtStep  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402

oracle = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
kernels = {}
for case in range(n_cases):
    dof, n_res, nprob = int(rng.integers(1, 13)), int(rng.integers(1, 31)), int(rng.integers(1, 20))
    J = rng.standard_normal((nprob, n_res, dof)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-2, 1.5))
    mode = int(rng.integers(0, 4))
    if mode == 1:
        J[: max(1, nprob // 3)] *= 1e-3  # the damping term dominates
    if mode == 2 and dof > 1:
        J[..., -1] = J[..., 0]  # rank deficient: only the damping makes the system definite
    e = rng.standard_normal((nprob, n_res)).astype(np.float32)
    jte = np.einsum("prd,pr->pd", J, e).astype(np.float32)
    lam = (10.0 ** rng.uniform(-6 if mode != 2 else -2, 3, nprob)).astype(np.float32)
    q_in = rng.uniform(-2, 2, (nprob, dof)).astype(np.float32)
    try:
        if (dof, n_res) not in kernels:
            kernels[(dof, n_res)] = LevenbergMarquardtStep.create_lm_warp_kernel(dof, n_res)
        q_out, pred = np.zeros((nprob, dof), np.float32), np.zeros(nprob, np.float32)
        wp.launch_tiled(kernels[(dof, n_res)], dim=[nprob], inputs=[wp.array(J), wp.array(jte), wp.array(lam), wp.array(q_in), wp.array(q_out), wp.array(pred)],
                        block_dim=32)
        q, p = oracle.lm_step(J, jte, lam, q_in)
        d_ref = q_out - q_in
        fin = np.isfinite(d_ref).all(-1)
        assert np.array_equal(np.isfinite(q - q_in).all(-1), fin), "finite pattern of the step"
        # (an ill-conditioned system amplifies the one-ulp differences of the Cholesky factor: bound relative to the step)
        np.testing.assert_allclose((q - q_in)[fin], d_ref[fin], rtol=0, atol=2e-5 * max(1e-12, float(np.abs(d_ref[fin]).max())), err_msg="delta")
        np.testing.assert_allclose(p[fin], pred[fin], rtol=0, atol=2e-5 * max(1e-12, float(np.abs(pred[fin]).max())), err_msg="pred_reduction")
    except AssertionError as ex:
        bad += 1
        print(f"FAILED case {case}: dof {dof} residuals {n_res} problems {nprob} mode {mode}: {str(ex)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
