"""The oracle against THE REFERENCE'S OWN CUDA KERNELS, executed on the CPU (``oracle/_ref/libcurobo_ref.so``).

The reference's kinematics, self-collision and B-spline kernels are CUDA C++; ``oracle/cuda_on_cpu`` compiles their
unmodified sources with g++ over a CUDA-on-CPU shim (a block = std::threads; barriers, warp shuffles, ballots and shared
memory as on the GPU) and ``oracle/ref_kernels.py`` launches them with the reference's launch geometry.  These tests put
the C oracle next to them on the same inputs: FK poses / spheres / Jacobian, B-spline and self collision agree TO THE LAST
BIT; the FK VJP and the centre of mass to a few ulp (the kernels sum over threads in tree order).

The library is built by ``__graft_entry__.build()`` where ``/root/reference`` exists and travels to the GPU box as a
prebuilt file; the tests skip where it is absent.  ``tests/golden/cuda_kernels_golden.npz`` keeps a set of these outputs
for places without the library (generator: ``tests/golden/make_cuda_kernels_golden.py``).
"""
import os

import numpy as np
import pytest

from conftest import load_model, sample_q

from oracle import ref_kernels

needs_ref = pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref/libcurobo_ref.so not built (no /root/reference here)")
ROBOTS = ["franka", "ur10e", "unitree_g1"]
KEYS = ("position", "velocity", "acceleration", "jerk")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cuda_kernels_golden.npz")


@pytest.fixture(scope="module")
def ref():
    return ref_kernels.ReferenceKernels()


@needs_ref
@pytest.mark.parametrize("robot", ROBOTS)
def test_fk_forward_kernels_bit_identical(robot, oracle, ref):
    model = load_model(robot)
    md = model.as_dict()
    q = sample_q(model, 21, seed=11)
    a = oracle.kinematics_forward(q, md, compute_jacobian=True, compute_com=True)
    b = ref.kinematics_forward(q, md, compute_jacobian=True, compute_com=True)           # spheres + Jacobian kernel
    c = ref.kinematics_forward(q, md, compute_spheres=False, compute_com=True)           # pose-only kernel
    for k in ("link_pos", "link_quat", "cumul_mat", "robot_spheres", "jacobian"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("link_pos", "link_quat", "cumul_mat"):
        assert np.array_equal(a[k], c[k]), k
    for other in (b, c):  # centre of mass: a 16- / 32-lane shuffle tree there, a loop here
        np.testing.assert_allclose(other["com"], a["com"], rtol=2e-6, atol=1e-6)


@needs_ref
def test_fk_sphere_sets_per_environment_bit_identical(oracle, ref):
    """two sphere sets (link_spheres [2, S, 4]) selected per trajectory through env_query_idx, horizon 3"""
    model = load_model("franka")
    md = dict(model.as_dict())
    rng = np.random.default_rng(8)
    base = np.asarray(md["link_spheres"], np.float32).reshape(-1, 4)
    other = base.copy()
    other[:, :3] += 0.01 * rng.standard_normal((base.shape[0], 3)).astype(np.float32)
    other[:, 3] = np.where(base[:, 3] > 0, base[:, 3] * 1.2, base[:, 3])
    md["link_spheres"] = np.stack([base, other])
    horizon, env = 3, np.array([1, 0, 0, 1], np.int32)
    q = sample_q(model, 4 * horizon, seed=15)
    a = oracle.kinematics_forward(q, md, horizon=horizon, env_query_idx=env)
    b = ref.kinematics_forward(q, md, horizon=horizon, env_query_idx=env)
    assert np.array_equal(a["robot_spheres"], b["robot_spheres"])
    assert not np.array_equal(a["robot_spheres"][0], oracle.kinematics_forward(q[:1], model.as_dict())["robot_spheres"][0])


@needs_ref
@pytest.mark.parametrize("robot", ROBOTS)
@pytest.mark.parametrize("with_com", [False, True])
def test_fk_backward_kernel(robot, with_com, oracle, ref):
    model = load_model(robot)
    md = model.as_dict()
    rng = np.random.default_rng(3)
    n = 9
    fk = oracle.kinematics_forward(sample_q(model, n, seed=12), md, compute_com=True)
    S, T = fk["robot_spheres"].shape[1], fk["link_pos"].shape[1]
    gs = rng.standard_normal((n, S, 4)).astype(np.float32)
    gs[..., 3] = 0
    gs[rng.random((n, S)) < 0.5] = 0
    gp, gq = rng.standard_normal((n, T, 3)).astype(np.float32), rng.standard_normal((n, T, 4)).astype(np.float32)
    gc = rng.standard_normal((n, 4)).astype(np.float32)
    args = (md, fk["cumul_mat"], gs, gp, gq) + ((gc, fk["com"]) if with_com else ())
    a, b = oracle.kinematics_backward(*args), ref.kinematics_backward(*args)
    np.testing.assert_allclose(b, a, rtol=0, atol=2e-6 * np.abs(a).max())


@needs_ref
@pytest.mark.parametrize("robot", ["franka", "ur10e"])
def test_self_collision_kernel_bit_identical(robot, oracle, ref):
    model = load_model(robot)
    sph = oracle.kinematics_forward(sample_q(model, 24, seed=13, scale=1.3), model.as_dict())["robot_spheres"]
    a = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.5)
    b = ref.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.5)
    assert (a["distance"] > 0).sum() >= 3
    assert np.array_equal(a["distance"], b["distance"]) and np.array_equal(a["gradient"], b["gradient"])
    assert np.array_equal(a["sparse_index"], b["sparse_index"])


@needs_ref
def test_self_collision_two_kernel_form_on_the_humanoid_pair_list(oracle, ref):
    """Unitree G1, 162 111 pairs: the reference splits them over blocks (self_collision_max_block_kernel) and reduces
    (self_collision_max_reduce_kernel); same distance, gradient and flags as the oracle"""
    model = load_model("unitree_g1")
    sph = oracle.kinematics_forward(sample_q(model, 4, seed=13, scale=1.2), model.as_dict())["robot_spheres"]
    a = oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.5)
    b = ref.self_collision_blocks(sph, model.sphere_padding, model.collision_pairs, 1.5, num_blocks_per_batch=10)
    assert model.collision_pairs.shape[0] > 160000 and (a["distance"] > 0).sum() >= 2
    assert np.array_equal(a["distance"], b["distance"]) and np.array_equal(a["gradient"], b["gradient"])
    assert np.array_equal(a["sparse_index"], b["sparse_index"])


@needs_ref
@pytest.mark.parametrize("robot", ["franka", "unitree_g1"])
def test_rnea_kernels(robot, oracle, ref):
    """rnea_forward_kernel / rnea_backward_kernel (one thread per element): torques, the forward cache (same layout as the
    oracle's: v, a, f per link) and the VJP w.r.t. q, qd, qdd"""
    model = load_model(robot)
    md = model.as_dict()
    rng = np.random.default_rng(7)
    q = sample_q(model, 12, seed=14)
    qd, qdd, gt = (rng.standard_normal(q.shape).astype(np.float32) for _ in range(3))
    ta, ca = oracle.rnea_forward(q, qd, qdd, md)
    tb, cb = ref.rnea_forward(q, qd, qdd, md)
    np.testing.assert_allclose(tb, ta, rtol=0, atol=1e-6 * np.abs(ta).max())
    np.testing.assert_allclose(cb, ca.reshape(q.shape[0], -1), rtol=0, atol=1e-6 * np.abs(ca).max())
    for a, b in zip(oracle.rnea_backward(gt, q, qd, ca, md), ref.rnea_backward(gt, q, qd, cb, md)):
        np.testing.assert_allclose(b, a, rtol=0, atol=3e-6 * np.abs(a).max())


def _bspline_case(degree, implicit):
    rng = np.random.default_rng(degree)
    b, nk, dof, interp = 7, 12, 7, 2
    ph = (nk + degree + 1) * interp + 1
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda n: {k: (rng.normal(size=(n, dof)) * 0.3).astype(np.float32) for k in KEYS}  # noqa: E731
    start, goal = mk(3), mk(2)
    sidx, gidx = rng.integers(0, 3, size=b).astype(np.int32), rng.integers(0, 2, size=b).astype(np.int32)
    dt, imp = np.array([0.05, 0.08], np.float32), np.array([implicit, implicit], np.uint8)
    g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
    return (u, start, goal, sidx, gidx, dt, imp, ph, degree), (*g, dt, gidx, imp, nk, degree)


@needs_ref
@pytest.mark.parametrize("degree", [3, 4, 5])
@pytest.mark.parametrize("implicit", [0, 1])
def test_bspline_kernels_bit_identical(degree, implicit, oracle, ref):
    fwd, bwd = _bspline_case(degree, implicit)
    a, b = oracle.bspline_forward(*fwd), ref.bspline_forward(*fwd)
    for k in KEYS + ("dt",):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(oracle.bspline_backward(*bwd), ref.bspline_backward(*bwd))


@needs_ref
@pytest.mark.parametrize("degree", [3, 4, 5])
def test_bspline_single_dt_kernel(degree, oracle, ref):
    """interpolate_bspline_single_dt_kernel (re-interpolation of solved knots at one dt, per-trajectory horizons, some beyond the
    output buffer): positions and dt identical, the derivatives to a few ulp of their range"""
    rng = np.random.default_rng(40 + degree)
    b, nk, dof, max_out = 6, 12, 7, 150
    u = rng.normal(size=(b, nk, dof)).astype(np.float32)
    mk = lambda n: {k: (rng.normal(size=(n, dof)) * 0.3).astype(np.float32) for k in KEYS}  # noqa: E731
    start, goal = mk(3), mk(2)
    sidx, gidx = rng.integers(0, 3, size=b).astype(np.int32), rng.integers(0, 2, size=b).astype(np.int32)
    horizons = rng.integers(20, 200, size=b).astype(np.int32)
    args = (u, start, goal, sidx, gidx, np.array([0.02], np.float32), np.array([0, 1], np.uint8), horizons, max_out, degree)
    a, c = oracle.bspline_single_dt(*args), ref.bspline_single_dt(*args)
    assert np.array_equal(a["position"], c["position"]) and np.array_equal(a["dt"], c["dt"])
    for k in KEYS[1:]:
        np.testing.assert_allclose(c[k], a[k], rtol=0, atol=2e-6 * max(1.0, np.abs(a[k]).max()), err_msg=k)


@needs_ref
@pytest.mark.parametrize("implicit_goal", [0, 1])
def test_legacy_position_kernels(implicit_goal, oracle, ref):
    """position_clique_loop_idx_fwd_kernel / _bwd_kernel (five-point stencils).  Under the implicit goal the forward pass replaces
    the last action by the goal, so its gradient is zero -- running the reference's kernel is what exposed a stencil expression
    the oracle (and the HIP kernel) had restated from a commented-out block there."""
    rng = np.random.default_rng(41)
    b, ah, dof = 7, 28, 7
    u = rng.normal(size=(b, ah, dof)).astype(np.float32)
    start = {k: (rng.normal(size=(3, dof)) * 0.3).astype(np.float32) for k in KEYS}
    goal = {"position": (rng.normal(size=(2, dof)) * 0.3).astype(np.float32), "velocity": np.zeros((2, dof), np.float32),
            "acceleration": np.zeros((2, dof), np.float32)}
    sidx, gidx = rng.integers(0, 3, size=b).astype(np.int32), rng.integers(0, 2, size=b).astype(np.int32)
    dt, imp = np.array([0.05, 0.08], np.float32), np.array([implicit_goal] * 2, np.uint8)
    a = oracle.differentiation_position_forward(u, start, goal["position"], sidx, gidx, dt, imp)
    c = ref.differentiation_position_forward(u, start, goal, sidx, gidx, dt, imp)
    assert np.array_equal(a["position"], c["position"]) and np.array_equal(a["dt"], c["dt"])
    for k in KEYS[1:]:
        np.testing.assert_allclose(c[k], a[k], rtol=0, atol=1e-6 * max(1.0, np.abs(a[k]).max()), err_msg=k)
    g = [rng.normal(size=(b, ah + 4, dof)).astype(np.float32) for _ in range(4)]
    ga, gc = oracle.differentiation_position_backward(*g, dt, gidx, imp), ref.differentiation_position_backward(*g, dt, gidx, imp)
    np.testing.assert_allclose(gc, ga, rtol=0, atol=1e-6 * np.abs(ga).max())
    if implicit_goal:
        assert np.all(ga[:, -1] == 0) and np.all(gc[:, -1] == 0)


@needs_ref
@pytest.mark.parametrize("use_rk2", [False, True])
def test_legacy_acceleration_kernels_bit_identical(use_rk2, oracle, ref):
    """acceleration_loop_idx_kernel and acceleration_loop_idx_rk2_kernel run the same semi-implicit Euler recursion (which is why
    the HIP entry point ignores use_rk2): both identical to the oracle"""
    rng = np.random.default_rng(9)
    b, H, dof = 5, 30, 7
    u = rng.normal(size=(b, H, dof)).astype(np.float32)
    start = {k: (rng.normal(size=(3, dof)) * 0.3).astype(np.float32) for k in ("position", "velocity", "acceleration")}
    sidx, dt = rng.integers(0, 3, size=b).astype(np.int32), (0.02 + 0.05 * rng.random(H)).astype(np.float32)
    a, c = oracle.integration_acceleration(u, start, sidx, dt), ref.integration_acceleration(u, start, sidx, dt, use_rk2=use_rk2)
    for k in a:
        assert np.array_equal(a[k], c[k]), k


def _ls_state(b, v, nls):
    z = np.zeros
    return dict(best_cost=np.full((b,), 1e9, np.float32), best_action=z((b, v), np.float32), best_iteration=z((b,), np.int16),
                current_iteration=z((b,), np.int16), converged=z((b,), np.uint8), exploration_cost=z((b,), np.float32),
                exploration_action=z((b, v), np.float32), exploration_gradient=z((b, v), np.float32), cost=z((b,), np.float32),
                action=z((b, v), np.float32), gradient=z((b, v), np.float32), exploration_idx=z((b, nls), np.int32),
                selected_idx=z((b, nls), np.int32))


@needs_ref
@pytest.mark.parametrize("shared_buffers", [False, True], ids=["kernel_lbfgs_step", "kernel_lbfgs_step_shared_memory"])
@pytest.mark.parametrize("b,v,m,stable", [(5, 84, 15, True), (3, 7, 5, True), (2, 175, 27, False)])
def test_lbfgs_step_kernels(b, v, m, stable, shared_buffers, oracle, ref):
    """kernel_lbfgs_step / kernel_lbfgs_step_shared_memory (lbfgs_step_kernel.cuh:18-199), run on the CPU, next to the oracle over
    m + 3 iterations of one optimiser state (the history fills up and rolls): history buffers identical, rho and the step to
    the rounding of the block reductions (the kernels sum the dot products over warps, the oracle in index order).  The first
    kernel's rho hand-over from warp 0 is spelled out by the build recipe (oracle/cuda_on_cpu/Makefile); v = 84 and 175 span
    several warps, v = 7 sits inside one."""
    rng = np.random.default_rng(b * 1000 + v)
    z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
    A = dict(step=z(b, v), rho=z(m, b), y=z(m, b, v), s=z(m, b, v), x0=z(b, v), g0=z(b, v))
    B = {k: a.copy() for k, a in A.items()}
    x = rng.normal(size=(b, v)).astype(np.float32)
    for it in range(m + 3):
        x = (x + 0.05 * rng.normal(size=(b, v))).astype(np.float32)
        g = (2.0 * x + 0.1 * rng.normal(size=(b, v))).astype(np.float32)  # gradient of a noisy bowl: y . s > 0 mostly
        if it == 4:
            g[0] = A["g0"][0] - (x[0] - A["x0"][0])  # one problem with y . s < 0: the stable-mode branches (rho = 0, gamma clamped)
        oracle.lbfgs_step(A["step"], A["rho"], A["y"], A["s"], x, g, A["x0"], A["g0"], 0.01, stable)
        ref.lbfgs_step(B["step"], B["rho"], B["y"], B["s"], x, g, B["x0"], B["g0"], 0.01, stable, shared_buffers=shared_buffers)
        for k in ("y", "s", "x0", "g0"):
            assert np.array_equal(A[k], B[k]), (it, k)
        np.testing.assert_allclose(B["rho"], A["rho"], rtol=2e-5, atol=1e-7 * np.abs(A["rho"]).max(), err_msg=f"iteration {it}")
        fin = np.isfinite(A["step"])
        assert np.array_equal(fin, np.isfinite(B["step"])), it
        scale = np.abs(A["step"][fin]).max()
        np.testing.assert_allclose(B["step"][fin], A["step"][fin], rtol=1e-3, atol=2e-5 * scale, err_msg=f"iteration {it}")
    assert np.abs(A["step"][np.isfinite(A["step"])]).max() > 0


@needs_ref
@pytest.mark.parametrize("name", ["ik", "trajopt"])
def test_lbfgs_step_kernel_against_the_reference_torch_twin(name, ref):
    """the CUDA kernel on the CPU against the golden of the reference's OWN torch twin (tests/golden/optim_golden.npz): the two
    reference implementations of the step agree here as they must on the GPU"""
    gold = np.load(os.path.join(os.path.dirname(GOLD), "optim_golden.npz"))
    q, g, step_ref = gold[f"lbfgs_{name}_q"], gold[f"lbfgs_{name}_g"], gold[f"lbfgs_{name}_step"]
    iters, b, v = q.shape
    m = gold[f"lbfgs_{name}_y"].shape[0]
    y, s, rho = np.zeros((m, b, v), np.float32), np.zeros((m, b, v), np.float32), np.zeros((m, b), np.float32)
    x0, g0 = gold[f"lbfgs_{name}_init_x0"].astype(np.float32).copy(), gold[f"lbfgs_{name}_init_g0"].astype(np.float32).copy()
    step = np.zeros((b, v), np.float32)
    for it in range(iters):
        ref.lbfgs_step(step, rho, y, s, np.ascontiguousarray(q[it]), np.ascontiguousarray(g[it]), x0, g0, 0.01, True)
        scale = np.abs(step_ref[it]).max()
        np.testing.assert_allclose(step, step_ref[it], atol=2e-4 * scale, rtol=2e-3, err_msg=f"iteration {it}")
    np.testing.assert_allclose(y, gold[f"lbfgs_{name}_y"], atol=1e-6)
    np.testing.assert_allclose(rho, gold[f"lbfgs_{name}_rho"], rtol=1e-4, atol=1e-6)


@needs_ref
@pytest.mark.parametrize("kind", ["wolfe", "strong_wolfe", "approx_wolfe"])
def test_line_search_kernel_identical_state(kind, oracle, ref):
    """four rounds of candidates through kernel_line_search and through the oracle: every state array identical (selected
    and exploration indices, best cost / action / iteration, the convergence flag)"""
    rng = np.random.default_rng(5)
    b, nls, v = 24, 4, 84
    sa, sb = _ls_state(b, v, nls), _ls_state(b, v, nls)
    picks = set()
    for _ in range(4):
        x, d = rng.normal(size=(b, nls, v)).astype(np.float32), rng.normal(size=(b, v)).astype(np.float32)
        c = (rng.random((b, nls)) * np.array([1, 0.8, 1.2, 2.0])).astype(np.float32)
        gx = (rng.normal(size=(b, nls, v)) * 0.3).astype(np.float32)
        al = np.array([0.0, 0.25, 0.5, 1.0], np.float32)
        for impl, st in ((oracle, sa), (ref, sb)):
            impl.line_search(st, c, x, gx, d, al, 1e-5, 0.9, kind == "strong_wolfe", kind == "approx_wolfe", 5, 0.0, 0.001)
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), k
        picks |= set(sa["selected_idx"][:, 0].tolist())
    assert len(picks) >= 3


def test_oracle_against_the_committed_outputs_of_the_reference_kernels(oracle):
    """the same comparison from a file, for machines without the library"""
    g = np.load(GOLD)
    model = load_model("franka")
    md = model.as_dict()
    a = oracle.kinematics_forward(g["fk/q"], md, compute_jacobian=True, compute_com=True)
    for k in ("link_pos", "link_quat", "cumul_mat", "robot_spheres", "jacobian"):
        assert np.array_equal(a[k], g["fk/" + k]), k
    np.testing.assert_allclose(a["com"], g["fk/com"], rtol=2e-6, atol=1e-6)
    vjp = oracle.kinematics_backward(md, g["fk/cumul_mat"], g["bwd/grad_spheres"], g["bwd/grad_link_pos"], g["bwd/grad_link_quat"])
    np.testing.assert_allclose(vjp, g["bwd/grad_q"], rtol=0, atol=2e-6 * np.abs(g["bwd/grad_q"]).max())
    sc = oracle.self_collision(g["self/spheres"], model.sphere_padding, model.collision_pairs, 1.5)
    assert np.array_equal(sc["distance"], g["self/distance"]) and np.array_equal(sc["gradient"], g["self/gradient"])
    assert np.array_equal(sc["sparse_index"], g["self/sparse_index"])
    fwd, bwd = _bspline_case(3, 1)
    bs = oracle.bspline_forward(*fwd)
    for k in KEYS:
        assert np.array_equal(bs[k], g["bspline/" + k]), k
    assert np.array_equal(oracle.bspline_backward(*bwd), g["bspline/grad_knots"])


@needs_ref
@pytest.mark.parametrize("interp", [1, 2, 4, 8, 16])
def test_bspline_backward_bit_identical_for_every_power_of_two_interpolation(interp, oracle, ref):
    """the reference adds the interpolation points' shares of a knot with a warp-segmented shuffle tree
    (bspline_gradient_util.cuh:83-105): strides interp / 2 .. 1, e.g. (v0 + v2) + (v1 + v3) at its default of four steps.
    The oracle follows that order, so the two agree to the last bit for every step count that divides the warp."""
    rng = np.random.default_rng(interp)
    degree, nk, dof, b = 3, 12, 7, 4
    ph = (nk + degree + 1) * interp + 1
    g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
    args = (*g, np.array([0.05], np.float32), np.zeros(b, np.int32), np.zeros(1, np.uint8), nk, degree)
    assert np.array_equal(oracle.bspline_backward(*args), ref.bspline_backward(*args))


@needs_ref
def test_reference_bspline_backward_is_not_a_sum_for_other_interpolation_steps(oracle, ref):
    """For 3, 5, 6 .. interpolation steps knots_per_warp * steps != 32 and the reference's shuffle tree pairs lanes of
    DIFFERENT knots: its result is off by O(1) (documented, not imitated: the oracle and the HIP kernels keep the sum, which
    the finite differences of the forward kernel confirm).  The reference's own configurations use 4 steps."""
    rng = np.random.default_rng(3)
    degree, nk, dof, b, interp = 3, 12, 7, 2, 3
    ph = (nk + degree + 1) * interp + 1
    g = [rng.normal(size=(b, ph, dof)).astype(np.float32) for _ in range(4)]
    dt, idx, imp = np.array([0.05], np.float32), np.zeros(b, np.int32), np.zeros(1, np.uint8)
    a, r = oracle.bspline_backward(*g, dt, idx, imp, nk, degree), ref.bspline_backward(*g, dt, idx, imp, nk, degree)
    assert np.abs(a - r).max() > 0.1 * np.abs(a).max()
    # the oracle's VJP is the adjoint of the (bit-identical) forward map: <J u, g> = <u, J^T g> on the knot-dependent part
    z = {k: np.zeros((1, dof), np.float32) for k in KEYS}
    u = rng.normal(size=(b, nk, dof)).astype(np.float64)
    f = lambda x: oracle.bspline_forward(x.astype(np.float32), z, z, idx, idx, dt, imp, ph, degree)  # noqa: E731
    f0, f1 = f(np.zeros_like(u)), f(u)
    lhs = sum(float(((f1[k].astype(np.float64) - f0[k]) * gk).sum()) for k, gk in zip(KEYS, g))
    rhs = float((u * a).sum())
    assert abs(lhs - rhs) < 2e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)


@needs_ref
def test_randomised_sweep_against_the_reference_kernels():
    """tests/randomised/sweep_reference_kernels.py at a small size: random batch sizes, joint ranges, gradients and spline shapes"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", "sweep_reference_kernels.py"), "16", "3"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]


@needs_ref
def test_reference_lbfgs_kernel_needs_as_many_threads_as_history_entries(oracle, ref):
    """The reference launches its L-BFGS step kernel with v_dim threads (optimization_config.py: threads_per_block = v_dim) and
    moves the rho buffer with ``threadIdx.x < history`` (lbfgs_step_kernel.cuh:162): with fewer variables than history
    entries -- a 6-dof arm under its shipped IK history of 7 -- part of the buffer is never loaded, and the kernel is not the
    algorithm of its own torch twin any more.  Documented, not imitated: the oracle (and the HIP kernel) agree with the twin
    for every v_dim (tests/randomised/sweep_reference_torch_optim.py runs v = 1, 3 under histories of up to 27).  (Through the
    CUDA-on-CPU shim other small shapes deviate as well -- v = 8 with any history above 1, v = 12 above 5 -- while v = 7 agrees
    up to a history of 7 and every v >= 32 tested agrees: docs/NOTEBOOK.md.)"""
    rng = np.random.default_rng(0)
    b, v, m = 3, 6, 7
    z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
    A = dict(step=z(b, v), rho=z(m, b), y=z(m, b, v), s=z(m, b, v), x0=z(b, v), g0=z(b, v))
    B = {k: a.copy() for k, a in A.items()}
    x = rng.normal(size=(b, v)).astype(np.float32)
    worst = 0.0
    for _ in range(m + 3):
        x = (x + 0.05 * rng.normal(size=(b, v))).astype(np.float32)
        g = (2.0 * x + 0.1 * rng.normal(size=(b, v))).astype(np.float32)
        oracle.lbfgs_step(A["step"], A["rho"], A["y"], A["s"], x, g, A["x0"], A["g0"], 0.01, True)
        ref.lbfgs_step(B["step"], B["rho"], B["y"], B["s"], x, g, B["x0"], B["g0"], 0.01, True)
        worst = max(worst, float(np.abs(A["rho"] - B["rho"]).max() / np.abs(A["rho"]).max()))
    assert worst > 0.1, worst  # an entry of the rho buffer the kernel never moved
    # one history entry fewer than variables: identical buffers (the regime the other tests and the sweep cover: v_dim >= 32, or 7)
    m = 5
    A = dict(step=z(b, v), rho=z(m, b), y=z(m, b, v), s=z(m, b, v), x0=z(b, v), g0=z(b, v))
    B = {k: a.copy() for k, a in A.items()}
    for _ in range(m + 3):
        x = (x + 0.05 * rng.normal(size=(b, v))).astype(np.float32)
        g = (2.0 * x + 0.1 * rng.normal(size=(b, v))).astype(np.float32)
        oracle.lbfgs_step(A["step"], A["rho"], A["y"], A["s"], x, g, A["x0"], A["g0"], 0.01, True)
        ref.lbfgs_step(B["step"], B["rho"], B["y"], B["s"], x, g, B["x0"], B["g0"], 0.01, True)
        np.testing.assert_allclose(B["rho"], A["rho"], rtol=2e-5, atol=1e-6 * np.abs(A["rho"]).max())
