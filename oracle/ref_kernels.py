"""ctypes wrapper of ``oracle/_ref/libcurobo_ref.so``: THE REFERENCE'S OWN CUDA KERNELS, run on the CPU.

Test infrastructure (like everything under ``oracle/``).  The library is built by ``make -C oracle/cuda_on_cpu`` (called
from ``__graft_entry__.build()`` when ``/root/reference`` exists) from the reference's kernel sources where they lie, with
g++, over a CUDA-on-CPU shim: a CUDA block is a group of cooperatively scheduled fibers, ``__syncthreads`` / ``__syncwarp`` /
``__shfl_*_sync`` / ``__ballot_sync`` / shared memory behave as on the GPU (``oracle/cuda_on_cpu/simt.cpp``).  Kernels:

    kinematics_forward_kernel, kinematics_forward_spheres_kernel, kinematics_forward_spheres_jacobian_kernel
    kinematics_backward_kernel                              (kernels/kinematics/*.cuh)
    self_collision_max_distance_kernel, self_collision_max_block_kernel + self_collision_max_reduce_kernel
                                                            (kernels/geometry/self_collision/*.cuh)
    interpolate_bspline_kernel, bspline_backward_kernel, interpolate_bspline_single_dt_kernel
                                                            (kernels/trajectory/bspline/*.cuh, degrees 3 / 4 / 5)
    position_clique_loop_idx_fwd_kernel / _bwd_kernel, acceleration_loop_idx_kernel / _rk2_kernel
                                                            (kernels/trajectory/legacy/*.cuh)
    kernel_line_search                                      (kernels/optimization/line_search/*.cuh)
    rnea_forward_kernel, rnea_backward_kernel               (kernels/dynamics/*.cuh; franka and unitree_g1 instantiations)

The methods mirror ``oracle.Oracle`` (same arguments, same result dictionaries) so that a test can put the two side by
side.  ``available()`` is False where the library was not built (no reference checkout and no prebuilt copy).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libcurobo_ref.so")


def available() -> bool:
    return os.path.exists(LIB_PATH)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


class ReferenceKernels:
    def __init__(self, path: str = LIB_PATH):
        self.lib = C.CDLL(path)

    # ------------------------------------------------------------------ kinematics
    def _tables(self, model):
        return dict(fixed=_f32(model["fixed_transforms"]), spheres=_f32(model["link_spheres"]), masses=_f32(model["link_masses_com"]),
                    jtype=np.ascontiguousarray(model["joint_map_type"], np.int8), jmap=_i16(model["joint_map"]), lmap=_i16(model["link_map"]),
                    tmap=_i16(model["tool_frame_map"]), smap=_i16(model["link_sphere_idx_map"]), lcd=_i16(model["link_chain_data"]),
                    lco=_i16(model["link_chain_offsets"]), jld=_i16(model["joint_links_data"]), jlo=_i16(model["joint_links_offsets"]),
                    jae=np.ascontiguousarray(model["joint_affects_endeffector"]).astype(np.bool_), joff=_f32(model["joint_offset_map"]))

    def kinematics_forward(self, q, model: Dict[str, np.ndarray], horizon: int = 1, env_query_idx: Optional[np.ndarray] = None,
                           compute_jacobian: bool = False, compute_com: bool = False, compute_spheres: bool = True):
        q = _f32(q).reshape(-1, q.shape[-1])
        n, d = q.shape
        t = self._tables(model)
        L, T = t["lmap"].shape[0], t["tmap"].shape[0]
        E, S = t["spheres"].shape[:2]
        env = np.ascontiguousarray(env_query_idx if env_query_idx is not None else np.zeros(max(n // max(horizon, 1), 1)), np.int32)
        out = {"link_pos": np.zeros((n, T, 3), np.float32), "link_quat": np.zeros((n, T, 4), np.float32),
               "cumul_mat": np.zeros((n, L, 3, 4), np.float32), "com": np.zeros((n, 4), np.float32) if compute_com else None,
               "robot_spheres": np.zeros((n, S if compute_spheres else 0, 4), np.float32),
               "jacobian": np.zeros((n, T, 6, d), np.float32) if compute_jacobian else None}
        com = out["com"] if compute_com else np.zeros((n, 4), np.float32)
        if not compute_spheres and not compute_jacobian:
            self.lib.ref_kinematics_forward(_p(out["link_pos"]), _p(out["link_quat"]), _p(com), _p(out["cumul_mat"]), _p(q), _p(t["fixed"]),
                                            _p(t["masses"]), _p(t["jtype"]), _p(t["jmap"]), _p(t["lmap"]), _p(t["tmap"]), _p(t["joff"]), n,
                                            horizon, L, d, T, int(compute_com))
            return out
        sph_out = out["robot_spheres"] if compute_spheres else np.zeros((n, S, 4), np.float32)
        self.lib.ref_kinematics_forward_spheres(
            _p(out["link_pos"]), _p(out["link_quat"]), _p(sph_out), _p(com), _p(out["jacobian"]), _p(out["cumul_mat"]), _p(q), _p(t["fixed"]),
            _p(t["spheres"]), _p(t["masses"]), _p(t["jtype"]), _p(t["jmap"]), _p(t["lmap"]), _p(t["tmap"]), _p(t["smap"]), _p(t["lcd"]),
            _p(t["lco"]), _p(t["jld"]), _p(t["jlo"]), _p(t["jae"]), _p(t["joff"]), _p(env), n, horizon, S, E, L, d, T, int(compute_com))
        return out

    def kinematics_backward(self, model, cumul_mat, grad_spheres, grad_link_pos=None, grad_link_quat=None, grad_com=None, batch_com=None,
                            horizon: int = 1, env_query_idx: Optional[np.ndarray] = None):
        t = self._tables(model)
        L, T = t["lmap"].shape[0], t["tmap"].shape[0]
        E, S = t["spheres"].shape[:2]
        cumul = _f32(cumul_mat).reshape(-1, L, 3, 4)
        n, d = cumul.shape[0], int(model["num_dof"])
        gs = _f32(grad_spheres).reshape(n, S, 4) if grad_spheres is not None else np.zeros((n, S, 4), np.float32)
        gp = _f32(grad_link_pos).reshape(n, T, 3) if grad_link_pos is not None else np.zeros((n, T, 3), np.float32)
        gq = _f32(grad_link_quat).reshape(n, T, 4) if grad_link_quat is not None else np.zeros((n, T, 4), np.float32)
        use_com = grad_com is not None and batch_com is not None
        gc = _f32(grad_com) if use_com else np.zeros((n, 4), np.float32)
        bc = _f32(batch_com) if use_com else np.zeros((n, 4), np.float32)
        env = np.ascontiguousarray(env_query_idx if env_query_idx is not None else np.zeros(max(n // max(horizon, 1), 1)), np.int32)
        out = np.zeros((n, d), np.float32)
        self.lib.ref_kinematics_backward(_p(out), _p(gp), _p(gq), _p(gs), _p(gc), _p(bc), _p(cumul), _p(t["spheres"]), _p(t["masses"]),
                                         _p(t["jtype"]), _p(t["jmap"]), _p(t["lmap"]), _p(t["tmap"]), _p(t["smap"]), _p(env), _p(t["lcd"]),
                                         _p(t["lco"]), _p(t["jld"]), _p(t["jlo"]), _p(t["jae"]), _p(t["joff"]), n, horizon, S, L, d, T, E,
                                         int(use_com))
        return out

    def kinematics_backward_jacobian(self, model, cumul_mat, grad_jacobian, grad_link_pos=None, grad_link_quat=None, grad_spheres=None,
                                     horizon: int = 1):
        """d/dq of <grad_jacobian, J(q)> (+ the pose / sphere terms when given): the COMPUTE_JACOBIAN_GRAD instantiation"""
        t = self._tables(model)
        L, T = t["lmap"].shape[0], t["tmap"].shape[0]
        E, S = t["spheres"].shape[:2]
        cumul = _f32(cumul_mat).reshape(-1, L, 3, 4)
        n, d = cumul.shape[0], int(model["num_dof"])
        gj = _f32(grad_jacobian).reshape(n, T, 6, d)
        gs = _f32(grad_spheres).reshape(n, S, 4) if grad_spheres is not None else np.zeros((n, S, 4), np.float32)
        gp = _f32(grad_link_pos).reshape(n, T, 3) if grad_link_pos is not None else np.zeros((n, T, 3), np.float32)
        gq = _f32(grad_link_quat).reshape(n, T, 4) if grad_link_quat is not None else np.zeros((n, T, 4), np.float32)
        env = np.zeros(max(n // max(horizon, 1), 1), np.int32)
        out = np.zeros((n, d), np.float32)
        self.lib.ref_kinematics_backward_jacobian(_p(out), _p(gp), _p(gq), _p(gs), _p(gj), _p(cumul), _p(t["spheres"]), _p(t["masses"]),
                                                  _p(t["jtype"]), _p(t["jmap"]), _p(t["lmap"]), _p(t["tmap"]), _p(t["smap"]), _p(env),
                                                  _p(t["lcd"]), _p(t["lco"]), _p(t["jld"]), _p(t["jlo"]), _p(t["jae"]), _p(t["joff"]), n,
                                                  horizon, S, L, d, T, E)
        return out

    # ------------------------------------------------------------------ self collision
    def self_collision(self, robot_spheres, sphere_padding, collision_pairs, weight: float, max_threads_per_block: int = 512,
                       write_grad: bool = True):
        rs = _f32(robot_spheres)
        S = rs.shape[-2]
        rs = rs.reshape(-1, S, 4)
        n = rs.shape[0]
        pairs = _i16(collision_pairs).reshape(-1, 2).copy()
        dist, grad = np.zeros(n, np.float32), np.zeros((n, S, 4), np.float32)
        flags, pd = np.zeros((n, S), np.uint8), np.zeros(1, np.float32)
        self.lib.ref_self_collision_distance(_p(dist), _p(grad), _p(pd), _p(flags), _p(rs), _p(_f32(sphere_padding)),
                                             _p(np.array([weight], np.float32)), _p(pairs), n, 1, S, pairs.shape[0], max_threads_per_block,
                                             int(write_grad))
        return {"distance": dist, "gradient": grad, "sparse_index": flags}

    def self_collision_blocks(self, robot_spheres, sphere_padding, collision_pairs, weight: float, num_blocks_per_batch: int,
                              max_threads_per_block: int = 512, write_grad: bool = True):
        """the two-kernel form the reference takes for long pair lists (max per block, then reduce)"""
        rs = _f32(robot_spheres)
        S = rs.shape[-2]
        rs = rs.reshape(-1, S, 4)
        n = rs.shape[0]
        pairs = _i16(collision_pairs).reshape(-1, 2).copy()
        dist, grad = np.zeros(n, np.float32), np.zeros((n, S, 4), np.float32)
        flags, pd = np.zeros((n, S), np.uint8), np.zeros(1, np.float32)
        bmv, bmi = np.zeros(n * num_blocks_per_batch, np.float32), np.zeros(2 * n * num_blocks_per_batch, np.int16)
        self.lib.ref_self_collision_distance_blocks(_p(dist), _p(grad), _p(pd), _p(flags), _p(rs), _p(_f32(sphere_padding)),
                                                    _p(np.array([weight], np.float32)), _p(pairs), _p(bmv), _p(bmi), num_blocks_per_batch,
                                                    n, 1, S, pairs.shape[0], max_threads_per_block, int(write_grad))
        return {"distance": dist, "gradient": grad, "sparse_index": flags}

    # ------------------------------------------------------------------ B-spline
    def bspline_forward(self, u, start, goal, start_idx, goal_idx, traj_dt, use_implicit_goal, padded_horizon: int, degree: int = 3):
        u = _f32(u)
        b, n_knots, dof = u.shape
        keys = ("position", "velocity", "acceleration", "jerk")
        outs = [np.zeros((b, padded_horizon, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros(b, np.float32)
        s, g = [_f32(start[k]) for k in keys], [_f32(goal[k]) for k in keys]
        rc = self.lib.ref_bspline_forward(*[_p(o) for o in outs], _p(out_dt), _p(u), *[_p(x) for x in s], *[_p(x) for x in g],
                                          _p(np.ascontiguousarray(start_idx, np.int32)), _p(np.ascontiguousarray(goal_idx, np.int32)),
                                          _p(_f32(traj_dt)), _p(np.ascontiguousarray(use_implicit_goal, np.uint8)), b, padded_horizon, dof,
                                          n_knots, degree)
        assert rc == 0
        return {**dict(zip(keys, outs)), "dt": out_dt}

    def bspline_backward(self, grad_p, grad_v, grad_a, grad_j, traj_dt, dt_idx, use_implicit_goal, n_knots: int, degree: int = 3):
        gp = _f32(grad_p)
        b, ph, dof = gp.shape
        out = np.zeros((b, n_knots, dof), np.float32)
        rc = self.lib.ref_bspline_backward(_p(out), _p(gp), _p(_f32(grad_v)), _p(_f32(grad_a)), _p(_f32(grad_j)), _p(_f32(traj_dt)),
                                           _p(np.ascontiguousarray(dt_idx, np.int32)), _p(np.ascontiguousarray(use_implicit_goal, np.uint8)),
                                           b, ph - 1, dof, n_knots, degree)
        assert rc == 0
        return out

    def bspline_single_dt(self, knots, start, goal, start_idx, goal_idx, interpolation_dt, use_implicit_goal, interpolation_horizon,
                          max_out_tsteps: int, degree: int = 3):
        u = _f32(knots)
        b, n_knots, dof = u.shape
        keys = ("position", "velocity", "acceleration", "jerk")
        outs = [np.zeros((b, max_out_tsteps, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros(b, np.float32)
        s, g = [_f32(start[k]) for k in keys], [_f32(goal[k]) for k in keys]
        rc = self.lib.ref_bspline_single_dt(*[_p(o) for o in outs], _p(out_dt), _p(u), _p(np.zeros((b, max(n_knots - 1, 1)), np.float32)),
                                            *[_p(x) for x in s], *[_p(x) for x in g], _p(np.ascontiguousarray(start_idx, np.int32)),
                                            _p(np.ascontiguousarray(goal_idx, np.int32)), _p(_f32(interpolation_dt)),
                                            _p(np.ascontiguousarray(use_implicit_goal, np.uint8)),
                                            _p(np.ascontiguousarray(interpolation_horizon, np.int32)), b, max_out_tsteps, dof, n_knots, degree)
        assert rc == 0
        return {**dict(zip(keys, outs)), "dt": out_dt}

    # ------------------------------------------------------------------ legacy POSITION control space
    def differentiation_position_forward(self, u_position, start, goal, start_idx, goal_idx, traj_dt, use_implicit_goal):
        """``goal``: dict with position / velocity / acceleration (the oracle's wrapper takes the goal position only: the kernel reads the
        other two only under the implicit goal, where they are zero by definition)"""
        u = _f32(u_position)
        b, ah, dof = u.shape
        horizon = ah + 4
        outs = [np.zeros((b, horizon, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros(b, np.float32)
        self.lib.ref_differentiation_position_forward(
            *[_p(o) for o in outs], _p(out_dt), _p(u), _p(_f32(start["position"])), _p(_f32(start["velocity"])),
            _p(_f32(start["acceleration"])), _p(_f32(goal["position"])), _p(_f32(goal["velocity"])), _p(_f32(goal["acceleration"])),
            _p(np.ascontiguousarray(start_idx, np.int32)), _p(np.ascontiguousarray(goal_idx, np.int32)), _p(_f32(traj_dt)),
            _p(np.ascontiguousarray(use_implicit_goal, np.uint8)), b, horizon, dof)
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2], "jerk": outs[3], "dt": out_dt}

    def differentiation_position_backward(self, grad_p, grad_v, grad_a, grad_j, traj_dt, dt_idx, use_implicit_goal):
        gp = _f32(grad_p)
        b, horizon, dof = gp.shape
        out = np.zeros((b, horizon - 4, dof), np.float32)
        self.lib.ref_differentiation_position_backward(_p(out), _p(gp), _p(_f32(grad_v)), _p(_f32(grad_a)), _p(_f32(grad_j)),
                                                       _p(_f32(traj_dt)), _p(np.ascontiguousarray(dt_idx, np.int32)),
                                                       _p(np.ascontiguousarray(use_implicit_goal, np.uint8)), b, horizon, dof)
        return out

    def integration_acceleration(self, u_acc, start, start_idx, traj_dt, use_rk2: bool = True):
        u = _f32(u_acc)
        b, horizon, dof = u.shape
        outs = [np.zeros((b, horizon, dof), np.float32) for _ in range(4)]
        rc = self.lib.ref_integration_acceleration(*[_p(o) for o in outs], _p(u), _p(_f32(start["position"])), _p(_f32(start["velocity"])),
                                                   _p(_f32(start["acceleration"])), _p(np.ascontiguousarray(start_idx, np.int32)),
                                                   _p(_f32(traj_dt)), b, horizon, dof, int(use_rk2))
        assert rc == 0, "horizon > 64 is not instantiated"
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2], "jerk": outs[3]}

    # ------------------------------------------------------------------ optimiser
    def lbfgs_step(self, step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0, epsilon: float, stable_mode: bool,
                   shared_buffers: bool = False):
        """in place like ``Oracle.lbfgs_step``: kernel_lbfgs_step<float, false, -1> (or its shared-memory form)"""
        m, b = y_buffer.shape[0], q.shape[0]
        v = int(np.prod(q.shape[1:]))
        for a in (step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        rc = self.lib.ref_lbfgs_step(_p(step_vec), _p(rho_buffer), _p(y_buffer), _p(s_buffer), _p(q), _p(x_0), _p(grad_0), _p(grad_q),
                                     C.c_float(epsilon), b, m, v, int(stable_mode), int(shared_buffers))
        assert rc == 0, "history must be 1..31 and the optimisation dimension <= 1024 (one thread each)"
        return step_vec

    def line_search(self, state, search_cost, search_action, search_gradient, step_direction, search_magnitudes, c_1: float, c_2: float,
                    strong_wolfe: bool, approx_wolfe: bool, convergence_iteration: int, cost_delta_threshold: float,
                    cost_relative_threshold: float):
        """in place on ``state``, like ``Oracle.line_search`` (kernel_line_search<float, -1>)"""
        b, nls = search_cost.shape[0], search_cost.shape[1]
        opt_dim = search_action.shape[-1]
        s = state
        self.lib.ref_line_search(_p(s["best_cost"]), _p(s["best_action"]), _p(s["best_iteration"]), _p(s["current_iteration"]),
                                 _p(s["converged"]), convergence_iteration, C.c_float(cost_delta_threshold),
                                 C.c_float(cost_relative_threshold), _p(s["exploration_cost"]), _p(s["exploration_action"]),
                                 _p(s["exploration_gradient"]), _p(s["exploration_idx"]), _p(s["cost"]), _p(s["action"]), _p(s["gradient"]),
                                 _p(s["selected_idx"]), _p(_f32(search_cost)), _p(_f32(search_action)), _p(_f32(search_gradient)),
                                 _p(_f32(step_direction)), _p(_f32(search_magnitudes)), C.c_float(c_1), C.c_float(c_2), int(strong_wolfe),
                                 int(approx_wolfe), nls, opt_dim, b)
        return state

    # ------------------------------------------------------------------ inverse dynamics
    def _rnea_tables(self, model):
        from .oracle import Oracle

        level_starts, level_links = Oracle.tree_levels(model["link_map"])
        return dict(fixed=_f32(model["fixed_transforms"]), masses=_f32(model["link_masses_com"]), inertias=_f32(model["link_inertias"]),
                    jtype=np.ascontiguousarray(model["joint_map_type"], np.int8), jmap=_i16(model["joint_map"]), lmap=_i16(model["link_map"]),
                    joff=_f32(model["joint_offset_map"]), starts=_i16(level_starts), links=_i16(level_links))

    def rnea_forward(self, q, qd, qdd, model, gravity=(0, 0, 0, 0, 0, 9.81)):
        """(tau [b, dof], the reference's opaque forward cache [b, links * 20]); rnea_forward_kernel<L, D, 1, false>"""
        q, qd, qdd = _f32(q), _f32(qd), _f32(qdd)
        b, dof = q.shape
        t = self._rnea_tables(model)
        L = t["fixed"].shape[0]
        tau, cache = np.zeros((b, dof), np.float32), np.zeros((b, L * 20), np.float32)
        rc = self.lib.ref_rnea_forward(_p(tau), _p(q), _p(qd), _p(qdd), _p(t["fixed"]), _p(t["masses"]), _p(t["inertias"]), _p(t["jtype"]),
                                       _p(t["jmap"]), _p(t["lmap"]), _p(t["joff"]), _p(_f32(gravity)), _p(t["starts"]), _p(t["links"]),
                                       _p(cache), b, len(t["starts"]) - 1, L, dof)
        if rc != 0:
            raise ValueError(f"rnea kernels are instantiated for franka (13 links / 7 dof) and unitree_g1 (56 / 49), not {L} / {dof}")
        return tau, cache

    def rnea_backward(self, grad_tau, q, qd, cache, model, gravity=(0, 0, 0, 0, 0, 9.81)):
        q, qd, gt = _f32(q), _f32(qd), _f32(grad_tau)
        b, dof = q.shape
        t = self._rnea_tables(model)
        L = t["fixed"].shape[0]
        g = [np.zeros((b, dof), np.float32) for _ in range(3)]
        rc = self.lib.ref_rnea_backward(_p(g[0]), _p(g[1]), _p(g[2]), _p(gt), _p(q), _p(qd), _p(t["fixed"]), _p(t["masses"]),
                                        _p(t["inertias"]), _p(t["jtype"]), _p(t["jmap"]), _p(t["lmap"]), _p(t["joff"]), _p(_f32(gravity)),
                                        _p(t["starts"]), _p(t["links"]), _p(_f32(cache)), b, len(t["starts"]) - 1, L, dof)
        if rc != 0:
            raise ValueError(f"rnea kernels are instantiated for franka and unitree_g1, not {L} / {dof}")
        return tuple(g)
