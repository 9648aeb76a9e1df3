"""The compile-time shape table of the fused rollout launch (csrc/fused_shapes.hpp) against the packaged robots: a robot whose
dimensions drift away from its row would silently lose the specialised kernel (the dispatch falls back to the generic one)."""

import re

import numpy as np
import pytest
import torch

from conftest import load_model


def _lane_len(model):
    from curobo_amd.backends.rollout import attach_self_lane_lists

    pairs = torch.as_tensor(np.asarray(model.collision_pairs)).to(torch.int16)
    attach_self_lane_lists(pairs, model.num_spheres)
    lists = getattr(pairs, "_self_lane_lists", None)  # (robots outside the lane form -- more than 128 spheres -- have none)
    return int(lists[1]) if lists is not None else 0


def _shape(model, padded_horizon, n_cub, n_vox, terms=False, kinds=None, plain_launch=True, **kw):
    from curobo_amd.backends.rollout import fused_shape_id

    d = model.as_dict()
    args = dict(padded_horizon=padded_horizon, n_knots=12, dof=model.num_dof, num_links=model.num_links, num_spheres=model.num_spheres,
                num_collision_pairs=len(model.collision_pairs), link_chain_len=len(d["link_chain_data"]), self_lane_len=_lane_len(model),
                max_cuboids=n_cub, max_voxel_grids=n_vox, kinds=kinds if kinds is not None else ((1 if n_cub else 0) | (2 if n_vox else 0)),
                with_trajopt_terms=terms, plain_launch=plain_launch)
    args.update(kw)
    return fused_shape_id(**args)


def test_packaged_robots_take_their_compile_time_shapes():
    franka, ur10e = load_model("franka"), load_model("ur10e")
    assert _shape(franka, 33, 4, 0) == 1, "BASELINE C2: Franka, 12 knots x 2, the four cuboid slots of the C2 world, plain launch"
    assert _shape(franka, 33, 3, 0) == 2 and _shape(franka, 33, 7, 0) == 2, "Franka, any cuboid scene, plain launch"
    assert _shape(franka, 33, 2, 1) == 2, "cuboids + ESDF"
    assert _shape(franka, 33, 4, 0, plain_launch=False) == 3, "materialised outputs / several environments / profile stamps"
    assert _shape(franka, 33, 4, 0, terms=True) == 2, "the full trajopt cost set as an optimiser iteration (pose + c-space STATE, no cost outputs)"
    assert _shape(franka, 33, 4, 0, terms=True, plain_launch=False) == 3, "... with metrics outputs / torque limits"
    assert _shape(franka, 65, 2, 1) == 4 and _shape(franka, 65, 5, 0, plain_launch=False) == 4, "BASELINE C5: horizon 64"
    assert _shape(ur10e, 33, 0, 1) == 5 and _shape(ur10e, 33, 4, 0) == 5, "BASELINE C3: UR10e"
    assert _shape(ur10e, 33, 0, 1, plain_launch=False) == 6


def test_anything_else_runs_the_generic_kernel():
    franka = load_model("franka")
    assert _shape(franka, 33, 4, 0, n_knots=10) == 0
    assert _shape(franka, 31, 4, 0) == 0
    assert _shape(franka, 33, 4, 0, self_lane_len=0) == 0, "without lane lists the pair pass walks pair_locations: not in the table"
    assert _shape(franka, 33, 4, 0, num_collision_pairs=0) == 0, "self collision off"
    assert _shape(franka, 33, 4, 0, bspline_degree=4) == 0 and _shape(franka, 33, 4, 0, sweep_steps=0) == 0
    assert _shape(franka, 33, 4, 0, kinds=7) == 0, "analytic primitives"
    assert _shape(load_model("unitree_g1"), 33, 4, 0) == 0


def test_table_rows_are_consistent_with_the_build():
    import os

    from curobo_amd.build import CSRC, compile_units, fused_shape_ids

    text = open(os.path.join(CSRC, "fused_shapes.hpp")).read()
    ids = [int(m) for m in re.findall(r"#define CUROBO_FUSED_SHAPE_(\d+) FusedShape<", text)]
    assert ids == fused_shape_ids() == list(range(1, len(ids) + 1))
    each = re.search(r"#define CUROBO_FUSED_FOR_EACH_SHAPE\(X\)(.*)", text).group(1)
    assert [int(m) for m in re.findall(r"X\((\d+)\)", each)] == ids
    for k in ids:
        assert re.search(rf"#define CUROBO_FUSED_SHAPE_{k}_KERNELS\(K\)", text)
        assert ("rollout_fused.hip", f"rollout_fused_shape{k}", [f"-DCUROBO_FUSED_SHAPE_TU={k}"]) in compile_units()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c2", "c2_materialized", "franka_3_cuboids", "franka_h65_mixed", "ur10e_esdf", "ur10e_esdf_materialized",
                                  "franka_terms", "franka_terms_metrics"])
def test_specialised_launch_is_bit_identical_to_the_generic_kernel(case, device):
    """the same launch through the compile-time shape and through the generic kernel: every output bit equal"""
    from curobo_amd.backends.rollout import set_fused_shapes_enabled
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, c3_voxel_world, c5_mixed_worlds, seed_knots, start_configuration

    robot = "ur10e" if case.startswith("ur10e_esdf") else "franka"
    materialize = case.endswith("_materialized")  # (a launch that writes positions / spheres is not the plain form: other shapes)
    model = load_model(robot)
    kin = KinematicsParams.from_model(model, device)
    interp = 4 if case == "franka_h65_mixed" else 2
    if case == "franka_3_cuboids":
        arrays = cuboid_scene_arrays([c2_world()[0][:3]])
    elif case == "franka_h65_mixed":
        arrays = c5_mixed_worlds(1, voxels=True)
    elif case.startswith("ur10e_esdf"):
        arrays = c3_voxel_world(64, 0.04)
    else:
        arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(arrays, device)
    B = 96
    knots = torch.as_tensor(seed_knots(model, B, 12, seed=5), device=device).reshape(B, -1)
    start = torch.as_tensor(start_configuration(model), device=device)
    if case.startswith("franka_terms"):  # the full cost set (pose + c-space + self + swept scene); _metrics: with the per-term outputs
        from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg

        rng = np.random.default_rng(2)
        ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(traj_dt=0.1))
        ro.update_start_state(start)
        gq = rng.normal(size=(3, 1, 1, 4)).astype(np.float32)
        gq /= np.linalg.norm(gq, axis=-1, keepdims=True)
        ro.update_goals(torch.as_tensor(rng.normal(size=(3, 1, 1, 3)).astype(np.float32) * 0.4, device=device), torch.as_tensor(gq, device=device),
                        torch.as_tensor(rng.integers(0, 3, size=B).astype(np.int32), device=device))
        assert ro.fused_available()
    else:
        ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(interpolation_steps=interp, fused_materialize=materialize))
        ro.update_start_state(start)
    outs = []
    try:
        for enabled in (True, False, True):
            set_fused_shapes_enabled(enabled)
            if case.startswith("franka_terms"):
                c, g = ro.cost_and_gradient_fused(knots, with_metrics=case.endswith("_metrics"))
            else:
                c, g = ro.cost_and_gradient(knots)
            torch.cuda.synchronize()
            outs.append((c.clone(), g.clone()))
    finally:
        set_fused_shapes_enabled(True)
    assert float(outs[0][0].abs().sum()) > 0
    for c, g in outs[1:]:
        assert torch.equal(c, outs[0][0]) and torch.equal(g, outs[0][1])


def test_run_time_shape_compiles_loads_and_serves_its_dimensions(tmp_path, monkeypatch):
    """backends/fused_jit: a shape the library does not ship (Franka with 10 knots: padded horizon 29) is compiled by hipcc,
    loaded, registered, and the dispatch query then answers with a run-time id (>= 100); the object is cached on disk (the second
    call does not compile); an object whose argument block has another size is refused.  hipcc cross-compiles without a GPU."""
    from curobo_amd._lib import load
    from curobo_amd.backends import fused_jit

    monkeypatch.setenv("CUROBO_HIP_JIT_CACHE", str(tmp_path))
    franka = load_model("franka")
    lane = _lane_len(franka)
    d = franka.as_dict()
    dims = dict(padded_horizon=29, n_knots=10, dof=franka.num_dof, num_links=franka.num_links, num_spheres=franka.num_spheres,
                num_collision_pairs=len(franka.collision_pairs), link_chain_len=len(d["link_chain_data"]), self_lane_len=lane)
    assert _shape(franka, 29, 4, 0, n_knots=10) == 0
    assert fused_jit.ensure_shape(**dims, num_obstacles=4, kernels=[(3, 3, 1, False)])
    built = sorted(p.name for p in tmp_path.iterdir() if p.suffix == ".so")
    assert len(built) == 2, "the plain form and the any-form shape"
    assert _shape(franka, 29, 4, 0, n_knots=10) >= 100 and _shape(franka, 29, 4, 0, n_knots=10, plain_launch=False) >= 100
    assert _shape(franka, 29, 4, 0, n_knots=10, kinds=2) == 0, "only the instantiations that were asked for"
    assert _shape(franka, 33, 4, 0) == 1 and _shape(franka, 31, 4, 0) == 0, "other dimensions are untouched"
    # cached: nothing is compiled again
    import subprocess

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("compiled again")))
    assert fused_jit.ensure_shape(**dims, num_obstacles=4, kernels=[(3, 3, 1, False)])
    spec = fused_jit.shape_spec(**dims, threads=512, plain=True)
    assert fused_jit.compile_shape(spec, [(3, 3, 1, False)]).endswith(built[0]) or fused_jit.compile_shape(spec, [(3, 3, 1, False)]).endswith(built[1])
    # an object built for another argument block is refused
    lib = load()
    assert lib.curobo_hip_rollout_fused_register_shape(None, 1) != 0
    import ctypes as C

    fn = C.cast(lib.curobo_hip_abi_version, C.c_void_p)
    assert lib.curobo_hip_rollout_fused_register_shape(fn, 12345) != 0 and b"argument block" in lib.curobo_hip_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("n_knots,steps", [(10, 2), (11, 3)])
def test_run_time_shape_is_bit_identical_to_the_generic_kernel(n_knots, steps, device, tmp_path, monkeypatch):
    """CollisionRolloutCfg(jit_shape=True) on dimensions outside the shipped table (Franka, 10 knots x 2; 11 knots x 3): the first
    call builds and registers the shape, later launches run it -- same bits as the generic kernel.  Three interpolation steps: the
    sample parameter h / 3 is where a shape (quotient folded at compile time) and the generic kernel (run-time division, this
    library is built without correctly rounded fp32 division) were one ulp apart until bspline_device.hpp::exact_ratio."""
    from curobo_amd.backends.rollout import set_fused_shapes_enabled
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    monkeypatch.setenv("CUROBO_HIP_JIT_CACHE", str(tmp_path))
    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    B = 64
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(n_knots=n_knots, interpolation_steps=steps, jit_shape=True))
    H = ro.cfg.padded_horizon
    assert H == (n_knots + 4) * steps + 1
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    x = torch.as_tensor(seed_knots(model, B, n_knots, seed=5), device=device).reshape(B, -1)
    assert _shape(model, H, 4, 0, n_knots=n_knots) == 0
    c1, g1 = [t.clone() for t in ro.cost_and_gradient(x)]  # (compiles here)
    assert _shape(model, H, 4, 0, n_knots=n_knots) >= 100
    c2, g2 = [t.clone() for t in ro.cost_and_gradient(x)]
    set_fused_shapes_enabled(False)
    try:
        c0, g0 = [t.clone() for t in ro.cost_and_gradient(x)]
    finally:
        set_fused_shapes_enabled(True)
    torch.cuda.synchronize()
    assert float(c0.abs().sum()) > 0
    assert torch.equal(c1, c0) and torch.equal(g1, g0) and torch.equal(c2, c0) and torch.equal(g2, g0)


@pytest.mark.gpu
def test_run_time_shape_of_the_full_trajopt_cost_set(device, tmp_path, monkeypatch):
    """TrajOptRolloutCfg(jit_shape=True), Franka with 8 knots: the optimiser-iteration form (pose + c-space STATE + self + swept
    scene, no cost outputs) and the metrics form each get their run-time shape; both give the generic kernel's bits."""
    from curobo_amd.backends.rollout import set_fused_shapes_enabled
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    monkeypatch.setenv("CUROBO_HIP_JIT_CACHE", str(tmp_path))
    model = load_model("franka")
    kin = KinematicsParams.from_model(model, device)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), device)
    B = 64
    ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(n_knots=8, traj_dt=0.1, jit_shape=True))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    rng = np.random.default_rng(4)
    gq = rng.normal(size=(2, 1, 1, 4)).astype(np.float32)
    gq /= np.linalg.norm(gq, axis=-1, keepdims=True)
    ro.update_goals(torch.as_tensor(rng.normal(size=(2, 1, 1, 3)).astype(np.float32) * 0.4, device=device), torch.as_tensor(gq, device=device),
                    torch.as_tensor(rng.integers(0, 2, size=B).astype(np.int32), device=device))
    x = torch.as_tensor(seed_knots(model, B, 8, seed=5), device=device).reshape(B, -1)
    assert ro.fused_available()
    H = ro.cfg.padded_horizon
    assert _shape(model, H, 4, 0, n_knots=8, terms=True) == 0
    outs = {}
    ro.cost_and_gradient(x)  # (the rollout's first call builds and registers both forms)
    for metrics in (False, True):
        c1, g1 = [t.clone() for t in ro.cost_and_gradient_fused(x.view(B, 8, -1), with_metrics=metrics)]
        set_fused_shapes_enabled(False)
        try:
            c0, g0 = [t.clone() for t in ro.cost_and_gradient_fused(x.view(B, 8, -1), with_metrics=metrics)]
        finally:
            set_fused_shapes_enabled(True)
        torch.cuda.synchronize()
        assert float(c0.abs().sum()) > 0 and torch.equal(c1, c0) and torch.equal(g1, g0)
        outs[metrics] = c0
    assert _shape(model, H, 4, 0, n_knots=8, terms=True) >= 100 and _shape(model, H, 4, 0, n_knots=8, terms=True, plain_launch=False) >= 100
    assert torch.equal(outs[False], outs[True]), "the outputs do not change the cost"
