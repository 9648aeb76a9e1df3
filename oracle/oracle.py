"""ctypes/numpy front-end of ``oracle/curobo_oracle.c`` (TEST INFRASTRUCTURE ONLY).

All inputs are numpy arrays; outputs are freshly allocated numpy arrays unless the reference
contract is stateful (self-collision gradient buffers, L-BFGS history, line-search state), in
which case the caller's arrays are updated in place exactly like the reference kernels do.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcurobo_oracle.so")
_SRC = os.path.join(_HERE, "curobo_oracle.c")


def build_oracle(force: bool = False) -> str:
    """Compile the C oracle with gcc (``make -C oracle``) if missing or stale."""
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB


_NATIVE_DIR = os.path.join(_HERE, "_native")


def build_native_oracle() -> str:
    """The same C source compiled FOR THE HOST IT RUNS ON (``gcc -O3 -march=native -fopenmp``, SURVEY.md
    section 8d) -- used only as the timed CPU baseline of bench.py, never as the checker (the checker
    build keeps -ffp-contract=off so its fp32 results do not depend on the host's FMA support).  Built
    into oracle/_native/ on the box that runs it (ignored by git and by gpurun: a -march=native object
    must not travel between machines)."""
    os.makedirs(_NATIVE_DIR, exist_ok=True)
    lib = os.path.join(_NATIVE_DIR, "libcurobo_oracle_native.so")
    if (not os.path.exists(lib)) or os.path.getmtime(lib) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-fopenmp", "-fvisibility=hidden",
                               "-shared", "-o", lib, _SRC, "-lm"])
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


class _SceneStruct(C.Structure):
    _fields_ = [
        ("cub_dims", C.c_void_p),
        ("cub_inv_pose", C.c_void_p),
        ("cub_enable", C.c_void_p),
        ("cub_count", C.c_void_p),
        ("max_cub", C.c_int),
        ("vox_params", C.c_void_p),
        ("vox_inv_pose", C.c_void_p),
        ("vox_enable", C.c_void_p),
        ("vox_count", C.c_void_p),
        ("vox_features", C.c_void_p),
        ("max_vox", C.c_int),
        ("vox_n_voxels", C.c_int),
        ("vox_max_distance", C.c_float),
        ("mesh_dims", C.c_void_p), ("mesh_inv_pose", C.c_void_p), ("mesh_enable", C.c_void_p), ("mesh_count", C.c_void_p),
        ("mesh_id", C.c_void_p), ("mesh_vertices", C.c_void_p), ("mesh_faces", C.c_void_p), ("mesh_vert_offset", C.c_void_p),
        ("mesh_face_offset", C.c_void_p), ("max_mesh", C.c_int),
    ]


def mesh_scene_arrays(envs) -> Dict[str, np.ndarray]:
    """``envs[e]`` = list of mesh obstacles as ``curobo_amd.scene.mesh.MeshStore`` takes them (name, vertices, faces, pose,
    scale, enable) -> the ``mesh_*`` arrays of the oracle's scene dictionary (pure numpy; no BVH)"""
    E = len(envs)
    n = max(1, max(len(e) for e in envs))
    cache, verts, faces, voff, foff = {}, [], [], [0], [0]
    mesh_id, dims = np.zeros((E, n), np.int32), np.zeros((E, n, 4), np.float32)
    inv_pose = np.zeros((E, n, 8), np.float32)
    inv_pose[..., 3] = 1.0
    enable, count = np.zeros((E, n), np.uint8), np.zeros((E,), np.int32)
    for e, obs in enumerate(envs):
        count[e] = len(obs)
        for i, o in enumerate(obs):
            key = o.get("mesh_name", o.get("name", f"mesh_{e}_{i}"))
            if key not in cache:
                v, f = np.asarray(o["vertices"], np.float32), np.asarray(o["faces"], np.int32)
                if o.get("scale") is not None:
                    v = v * np.asarray(o["scale"], np.float32).reshape(1, 3)
                cache[key] = len(voff) - 1
                verts.append(v)
                faces.append(f)
                voff.append(voff[-1] + len(v))
                foff.append(foff[-1] + len(f))
            m = cache[key]
            used = verts[m][faces[m].reshape(-1)]
            mesh_id[e, i] = m
            dims[e, i, :3] = used.max(0) - used.min(0)
            p = np.asarray(o["pose"], np.float64)
            q = p[3:7] / np.linalg.norm(p[3:7])
            qi = np.array([q[0], -q[1], -q[2], -q[3]])
            t = 2.0 * np.cross(qi[1:], p[:3])
            inv_pose[e, i, :3] = -(p[:3] + qi[0] * t + np.cross(qi[1:], t))
            inv_pose[e, i, 3:7] = qi
            enable[e, i] = 1 if o.get("enable", True) else 0
    return {"mesh_id": mesh_id, "mesh_dims": dims, "mesh_inv_pose": inv_pose, "mesh_enable": enable, "mesh_count": count,
            "mesh_vertices": np.concatenate(verts).astype(np.float32), "mesh_faces": np.concatenate(faces).astype(np.int32),
            "mesh_vert_offset": np.asarray(voff, np.int32), "mesh_face_offset": np.asarray(foff, np.int32)}


class Oracle:
    """Thin wrapper; one instance per process is enough (``load_oracle()``)."""

    def __init__(self, path: Optional[str] = None):
        self.lib = C.CDLL(path or build_oracle())
        self.lib.orc_num_threads.restype = C.c_int
        self.lib.orc_get_frame_arithmetic.restype = C.c_int
        for name in (
            "orc_kinematics_forward",
            "orc_kinematics_backward",
            "orc_self_collision",
            "orc_scene_collision",
            "orc_bspline_forward",
            "orc_bspline_backward",
            "orc_bspline_single_dt",
            "orc_differentiation_position_forward",
            "orc_differentiation_position_backward",
            "orc_integration_acceleration",
            "orc_rnea_forward",
            "orc_rnea_backward",
            "orc_lm_step",
            "orc_lbfgs_step",
            "orc_line_search",
            "orc_trajectory_cost_sum",
            "orc_tool_pose_distance",
            "orc_cspace_position_cost",
            "orc_cspace_state_cost",
            "orc_set_num_threads",
            "orc_set_frame_arithmetic",
            "orc_set_mesh_sign_rule",
        ):
            getattr(self.lib, name).restype = None

    # ------------------------------------------------------------------ threads
    def num_threads(self) -> int:
        return int(self.lib.orc_num_threads())

    def set_num_threads(self, n: int) -> None:
        self.lib.orc_set_num_threads(C.c_int(n))

    def set_frame_arithmetic(self, mode: str) -> None:
        """"reference" (Warp's quat_rotate, the default) or "device" (the rotation-matrix fma form of the HIP path) for the
        world -> obstacle-frame transform of the scene-collision stage; see orc_set_frame_arithmetic in curobo_oracle.c."""
        self.lib.orc_set_frame_arithmetic(C.c_int({"reference": 0, "device": 1}[mode]))

    MESH_SIGN_RULES = {"winding": 0, "rays": 1}

    def set_mesh_sign_rule(self, rule: str) -> None:
        """"winding" (generalised winding number, the default) or "rays" (Warp's mesh_query_point as published: +x / +y / +z
        rays, inside iff every ray's nearest hit is a back face) for the sign of mesh queries; see curobo_oracle.c."""
        self.lib.orc_set_mesh_sign_rule(C.c_int(self.MESH_SIGN_RULES[rule]))

    def mesh_sign_rule(self) -> str:
        return {v: k for k, v in self.MESH_SIGN_RULES.items()}[int(self.lib.orc_get_mesh_sign_rule())]

    def frame_arithmetic(self) -> str:
        return ("reference", "device")[int(self.lib.orc_get_frame_arithmetic())]

    # ------------------------------------------------------------------ FK
    def kinematics_forward(
        self,
        q: np.ndarray,
        model: Dict[str, np.ndarray],
        horizon: int = 1,
        env_query_idx: Optional[np.ndarray] = None,
        compute_jacobian: bool = False,
        compute_com: bool = False,
        compute_spheres: bool = True,
    ) -> Dict[str, np.ndarray]:
        q = _f32(q).reshape(-1, q.shape[-1])
        n, d = q.shape
        L = model["link_map"].shape[0]
        T = model["tool_frame_map"].shape[0]
        spheres = _f32(model["link_spheres"])
        num_envs = spheres.shape[0]
        S = spheres.shape[1] if compute_spheres else 0
        if env_query_idx is None:
            env_query_idx = np.zeros(max(n // max(horizon, 1), 1), dtype=np.int32)
        env_query_idx = np.ascontiguousarray(env_query_idx, dtype=np.int32)
        out = {
            "link_pos": np.zeros((n, T, 3), np.float32),
            "link_quat": np.zeros((n, T, 4), np.float32),
            "robot_spheres": np.zeros((n, max(S, 0), 4), np.float32),
            "cumul_mat": np.zeros((n, L, 3, 4), np.float32),
            "com": np.zeros((n, 4), np.float32) if compute_com else None,
            "jacobian": np.zeros((n, T, 6, d), np.float32) if compute_jacobian else None,
        }
        jae = np.ascontiguousarray(model["joint_affects_endeffector"]).astype(np.uint8)
        self.lib.orc_kinematics_forward(
            _ptr(out["link_pos"]), _ptr(out["link_quat"]),
            _ptr(out["robot_spheres"]) if S > 0 else None,
            _ptr(out["com"]), _ptr(out["jacobian"]), _ptr(out["cumul_mat"]),
            _ptr(q), _ptr(_f32(model["fixed_transforms"])), _ptr(spheres),
            _ptr(_f32(model["link_masses_com"])),
            _ptr(np.ascontiguousarray(model["joint_map_type"], np.int8)),
            _ptr(np.ascontiguousarray(model["joint_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["tool_frame_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_sphere_idx_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_chain_data"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_chain_offsets"], np.int16)),
            _ptr(np.ascontiguousarray(model["joint_links_data"], np.int16)),
            _ptr(np.ascontiguousarray(model["joint_links_offsets"], np.int16)),
            _ptr(jae), _ptr(_f32(model["joint_offset_map"])), _ptr(env_query_idx),
            C.c_int(n), C.c_int(horizon), C.c_int(S), C.c_int(num_envs), C.c_int(L), C.c_int(d),
            C.c_int(T),
        )
        return out

    def kinematics_backward(
        self,
        model: Dict[str, np.ndarray],
        cumul_mat: np.ndarray,
        grad_spheres: Optional[np.ndarray],
        grad_link_pos: Optional[np.ndarray] = None,
        grad_link_quat: Optional[np.ndarray] = None,
        grad_com: Optional[np.ndarray] = None,
        batch_com: Optional[np.ndarray] = None,
        horizon: int = 1,
        env_query_idx: Optional[np.ndarray] = None,
    ) -> np.ndarray:
        L = model["link_map"].shape[0]
        T = model["tool_frame_map"].shape[0]
        cumul = _f32(cumul_mat).reshape(-1, L, 3, 4)
        n = cumul.shape[0]
        d = int(model["num_dof"])
        spheres = _f32(model["link_spheres"])
        num_envs, S = spheres.shape[0], spheres.shape[1]
        if grad_spheres is None:
            S = 0
        else:
            grad_spheres = _f32(grad_spheres).reshape(n, S, 4)
        gp = _f32(grad_link_pos).reshape(n, T, 3) if grad_link_pos is not None else np.zeros((n, T, 3), np.float32)
        gq = _f32(grad_link_quat).reshape(n, T, 4) if grad_link_quat is not None else np.zeros((n, T, 4), np.float32)
        if env_query_idx is None:
            env_query_idx = np.zeros(max(n // max(horizon, 1), 1), dtype=np.int32)
        env_query_idx = np.ascontiguousarray(env_query_idx, dtype=np.int32)
        out = np.zeros((n, d), np.float32)
        compute_com = grad_com is not None and batch_com is not None
        self.lib.orc_kinematics_backward(
            _ptr(out), _ptr(gp), _ptr(gq), _ptr(grad_spheres),
            _ptr(_f32(grad_com)) if compute_com else None,
            _ptr(_f32(batch_com)) if compute_com else None,
            _ptr(cumul), _ptr(spheres), _ptr(_f32(model["link_masses_com"])),
            _ptr(np.ascontiguousarray(model["joint_map_type"], np.int8)),
            _ptr(np.ascontiguousarray(model["joint_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["tool_frame_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_sphere_idx_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_chain_data"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_chain_offsets"], np.int16)),
            _ptr(_f32(model["joint_offset_map"])), _ptr(env_query_idx),
            C.c_int(n), C.c_int(horizon), C.c_int(S), C.c_int(num_envs), C.c_int(L), C.c_int(d),
            C.c_int(T), C.c_int(1 if compute_com else 0),
        )
        return out

    # ------------------------------------------------------------------ self collision
    def self_collision(
        self,
        robot_spheres: np.ndarray,
        sphere_padding: np.ndarray,
        collision_pairs: np.ndarray,
        weight: float,
        out_gradient: Optional[np.ndarray] = None,
        sparse_index: Optional[np.ndarray] = None,
        store_pair_distance: bool = False,
        write_grad: bool = True,
    ) -> Dict[str, np.ndarray]:
        rs = _f32(robot_spheres)
        S = rs.shape[-2]
        rs = rs.reshape(-1, S, 4)
        n = rs.shape[0]
        pairs = np.ascontiguousarray(collision_pairs, np.int16).reshape(-1, 2)
        P = pairs.shape[0]
        if out_gradient is None:
            out_gradient = np.zeros((n, S, 4), np.float32)
        if sparse_index is None:
            sparse_index = np.zeros((n, S), np.uint8)
        dist = np.zeros((n,), np.float32)
        pair_idx = np.zeros((n, 2), np.int16)
        pair_d = np.zeros((n, P), np.float32) if store_pair_distance else None
        w = np.array([weight], np.float32)
        self.lib.orc_self_collision(
            _ptr(dist), _ptr(out_gradient), _ptr(pair_d), _ptr(sparse_index), _ptr(pair_idx),
            _ptr(rs), _ptr(_f32(sphere_padding)), _ptr(w), _ptr(pairs),
            C.c_int(n), C.c_int(S), C.c_int(P), C.c_int(int(store_pair_distance)),
            C.c_int(int(write_grad)),
        )
        return {
            "distance": dist,
            "gradient": out_gradient,
            "sparse_index": sparse_index,
            "pair_idx": pair_idx,
            "pair_distance": pair_d,
        }

    # ------------------------------------------------------------------ scene collision
    def scene_collision(
        self,
        spheres: np.ndarray,
        scene: Dict[str, np.ndarray],
        weight: float,
        activation_distance: float,
        env_query_idx: Optional[np.ndarray] = None,
        use_multi_env: bool = False,
        sweep: bool = False,
        enable_speed_metric: bool = False,
        speed_dt: float = 0.02,
    ) -> Dict[str, np.ndarray]:
        sp = _f32(spheres)
        assert sp.ndim == 4, "spheres must be [batch, horizon, num_spheres, 4]"
        b, h, S, _ = sp.shape
        st = _SceneStruct()
        keep = []

        def hold(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        if scene.get("cuboid_dims") is not None and scene["cuboid_dims"].size > 0:
            dims = scene["cuboid_dims"]
            st.max_cub = int(dims.shape[1])
            st.cub_dims = hold(dims, np.float32)
            st.cub_inv_pose = hold(scene["cuboid_inv_pose"], np.float32)
            st.cub_enable = hold(scene["cuboid_enable"], np.uint8)
            st.cub_count = hold(scene["cuboid_count"], np.int32)
        else:
            st.max_cub = 0
        if scene.get("voxel_params") is not None and scene["voxel_params"].size > 0:
            prm = scene["voxel_params"]
            st.max_vox = int(prm.shape[1])
            st.vox_params = hold(prm, np.float32)
            st.vox_inv_pose = hold(scene["voxel_inv_pose"], np.float32)
            st.vox_enable = hold(scene["voxel_enable"], np.uint8)
            st.vox_count = hold(scene["voxel_count"], np.int32)
            feats = np.ascontiguousarray(scene["voxel_features"], dtype=np.float16)
            keep.append(feats)
            st.vox_features = feats.view(np.uint16).ctypes.data
            st.vox_n_voxels = int(feats.reshape(prm.shape[0], prm.shape[1], -1).shape[-1])
            st.vox_max_distance = float(scene.get("voxel_max_distance", 10000.0))
        else:
            st.max_vox = 0
        if scene.get("mesh_id") is not None and scene["mesh_id"].size > 0:
            st.max_mesh = int(scene["mesh_id"].shape[1])
            st.mesh_id = hold(scene["mesh_id"], np.int32)
            st.mesh_dims = hold(scene["mesh_dims"], np.float32)
            st.mesh_inv_pose = hold(scene["mesh_inv_pose"], np.float32)
            st.mesh_enable = hold(scene["mesh_enable"], np.uint8)
            st.mesh_count = hold(scene["mesh_count"], np.int32)
            st.mesh_vertices = hold(scene["mesh_vertices"], np.float32)
            st.mesh_faces = hold(scene["mesh_faces"], np.int32)
            st.mesh_vert_offset = hold(scene["mesh_vert_offset"], np.int32)
            st.mesh_face_offset = hold(scene["mesh_face_offset"], np.int32)
        else:
            st.max_mesh = 0
        if env_query_idx is None:
            env_query_idx = np.zeros((b,), np.int32)
        env_query_idx = np.ascontiguousarray(env_query_idx, np.int32)
        dist = np.zeros((b, h, S), np.float32)
        grad = np.zeros((b, h, S, 4), np.float32)
        w = np.array([weight], np.float32)
        eta = np.array([activation_distance], np.float32)
        dt = np.array([speed_dt], np.float32)
        self.lib.orc_scene_collision(
            _ptr(dist), _ptr(grad), _ptr(sp), C.byref(st), _ptr(w), _ptr(eta), _ptr(env_query_idx),
            C.c_int(b), C.c_int(h), C.c_int(S), C.c_int(int(use_multi_env)),
            C.c_int(3 if sweep else 0), C.c_int(int(enable_speed_metric)), _ptr(dt),
        )
        return {"distance": dist, "gradient": grad}

    def mesh_query(self, points, vertices, faces, max_distance: float):
        """points [n, 3] in the mesh frame -> (sdf [n], gradient [n, 3]): brute force over every triangle, winding-number sign"""
        p = _f32(points).reshape(-1, 3)
        v, f = _f32(vertices), np.ascontiguousarray(faces, np.int32)
        sdf, grad = np.zeros(p.shape[0], np.float32), np.zeros((p.shape[0], 3), np.float32)
        self.lib.orc_mesh_query(_ptr(sdf), _ptr(grad), _ptr(p), _ptr(v), _ptr(f), C.c_int(f.shape[0]), C.c_float(max_distance),
                                C.c_int(p.shape[0]))
        return sdf, grad

    # ------------------------------------------------------------------ B-spline
    def bspline_forward(self, u, start, goal, start_idx, goal_idx, traj_dt, use_implicit_goal,
                        padded_horizon: int, degree: int = 3):
        """start/goal: dict with position/velocity/acceleration/jerk arrays [n_states, dof]."""
        u = _f32(u)
        b, n_knots, dof = u.shape
        outs = [np.zeros((b, padded_horizon, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros((b,), np.float32)
        keys = ("position", "velocity", "acceleration", "jerk")
        s = [_f32(start[k]) for k in keys]
        g = [_f32(goal[k]) for k in keys]
        self.lib.orc_bspline_forward(
            *[_ptr(o) for o in outs], _ptr(out_dt), _ptr(u), *[_ptr(x) for x in s],
            *[_ptr(x) for x in g], _ptr(np.ascontiguousarray(start_idx, np.int32)),
            _ptr(np.ascontiguousarray(goal_idx, np.int32)), _ptr(_f32(traj_dt)),
            _ptr(np.ascontiguousarray(use_implicit_goal, np.uint8)),
            C.c_int(b), C.c_int(padded_horizon), C.c_int(dof), C.c_int(n_knots), C.c_int(degree),
        )
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2],
                "jerk": outs[3], "dt": out_dt}

    def bspline_single_dt(self, knots, start, goal, start_idx, goal_idx, interpolation_dt, use_implicit_goal,
                          interpolation_horizon, max_out_tsteps: int, degree: int = 3):
        """reference launch_bspline_interpolation_single_dt_kernel: one dt, per-trajectory horizons."""
        u = _f32(knots)
        b, n_knots, dof = u.shape
        outs = [np.zeros((b, max_out_tsteps, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros((b,), np.float32)
        keys = ("position", "velocity", "acceleration", "jerk")
        s = [_f32(start[k]) for k in keys]
        g = [_f32(goal[k]) for k in keys]
        self.lib.orc_bspline_single_dt(
            *[_ptr(o) for o in outs], _ptr(out_dt), _ptr(u), *[_ptr(x) for x in s], *[_ptr(x) for x in g],
            _ptr(np.ascontiguousarray(start_idx, np.int32)), _ptr(np.ascontiguousarray(goal_idx, np.int32)),
            _ptr(_f32(interpolation_dt)), _ptr(np.ascontiguousarray(use_implicit_goal, np.uint8)),
            _ptr(np.ascontiguousarray(interpolation_horizon, np.int32)),
            C.c_int(b), C.c_int(max_out_tsteps), C.c_int(dof), C.c_int(n_knots), C.c_int(degree),
        )
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2], "jerk": outs[3], "dt": out_dt}

    def bspline_backward(self, grad_p, grad_v, grad_a, grad_j, traj_dt, dt_idx, use_implicit_goal,
                         n_knots: int, degree: int = 3):
        gp = _f32(grad_p)
        b, ph, dof = gp.shape
        out = np.zeros((b, n_knots, dof), np.float32)
        self.lib.orc_bspline_backward(
            _ptr(out), _ptr(gp), _ptr(_f32(grad_v)), _ptr(_f32(grad_a)), _ptr(_f32(grad_j)),
            _ptr(_f32(traj_dt)), _ptr(np.ascontiguousarray(dt_idx, np.int32)),
            _ptr(np.ascontiguousarray(use_implicit_goal, np.uint8)),
            C.c_int(b), C.c_int(ph), C.c_int(dof), C.c_int(n_knots), C.c_int(degree),
        )
        return out

    # ------------------------------------------------------------------ legacy transitions
    def differentiation_position_forward(self, u_position, start, goal_position, start_idx, goal_idx, traj_dt,
                                         use_implicit_goal):
        """reference launch_differentiation_position_forward_kernel; u [b, horizon-4, dof] -> 4 x [b, horizon, dof]."""
        u = _f32(u_position)
        b, ah, dof = u.shape
        horizon = ah + 4
        outs = [np.zeros((b, horizon, dof), np.float32) for _ in range(4)]
        out_dt = np.zeros((b,), np.float32)
        self.lib.orc_differentiation_position_forward(
            *[_ptr(o) for o in outs], _ptr(out_dt), _ptr(u), _ptr(_f32(start["position"])), _ptr(_f32(start["velocity"])),
            _ptr(_f32(start["acceleration"])), _ptr(_f32(goal_position)), _ptr(np.ascontiguousarray(start_idx, np.int32)),
            _ptr(np.ascontiguousarray(goal_idx, np.int32)), _ptr(_f32(traj_dt)),
            _ptr(np.ascontiguousarray(use_implicit_goal, np.uint8)), C.c_int(b), C.c_int(horizon), C.c_int(dof))
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2], "jerk": outs[3], "dt": out_dt}

    def differentiation_position_backward(self, grad_p, grad_v, grad_a, grad_j, traj_dt, dt_idx, use_implicit_goal):
        gp = _f32(grad_p)
        b, horizon, dof = gp.shape
        out = np.zeros((b, horizon - 4, dof), np.float32)
        self.lib.orc_differentiation_position_backward(
            _ptr(out), _ptr(gp), _ptr(_f32(grad_v)), _ptr(_f32(grad_a)), _ptr(_f32(grad_j)), _ptr(_f32(traj_dt)),
            _ptr(np.ascontiguousarray(dt_idx, np.int32)), _ptr(np.ascontiguousarray(use_implicit_goal, np.uint8)),
            C.c_int(b), C.c_int(horizon), C.c_int(dof))
        return out

    def integration_acceleration(self, u_acc, start, start_idx, traj_dt):
        u = _f32(u_acc)
        b, horizon, dof = u.shape
        outs = [np.zeros((b, horizon, dof), np.float32) for _ in range(4)]
        self.lib.orc_integration_acceleration(
            *[_ptr(o) for o in outs], _ptr(u), _ptr(_f32(start["position"])), _ptr(_f32(start["velocity"])),
            _ptr(_f32(start["acceleration"])), _ptr(np.ascontiguousarray(start_idx, np.int32)), _ptr(_f32(traj_dt)),
            C.c_int(b), C.c_int(horizon), C.c_int(dof))
        return {"position": outs[0], "velocity": outs[1], "acceleration": outs[2], "jerk": outs[3]}

    # ------------------------------------------------------------------ inverse dynamics
    @staticmethod
    def tree_levels(link_map):
        """(level_starts, level_links): links grouped by tree depth, reference
        KinematicsParams._compute_link_levels (robot/types/kinematics_params.py:255-290)."""
        link_map = np.asarray(link_map)
        n = len(link_map)
        depth = np.zeros(n, np.int64)
        for k in range(n):
            p = int(link_map[k])
            depth[k] = 0 if (p < 0 or p == k) else depth[p] + 1
        order = np.argsort(depth, kind="stable")
        starts = np.searchsorted(depth[order], np.arange(depth.max() + 2))
        return starts.astype(np.int16), order.astype(np.int16)

    def rnea_forward(self, q, qd, qdd, model, gravity=(0, 0, 0, 0, 0, 9.81), f_ext=None):
        """tau[b, dof], cache[b, L, 20] = (v6, a6, f6, 0, 0) per link."""
        q, qd, qdd = _f32(q), _f32(qd), _f32(qdd)
        b, dof = q.shape
        L = model["fixed_transforms"].shape[0]
        tau = np.zeros((b, dof), np.float32)
        cache = np.zeros((b, L, 20), np.float32)
        _, level_links = self.tree_levels(model["link_map"])
        fe = None if f_ext is None else _f32(f_ext)
        self.lib.orc_rnea_forward(
            _ptr(tau), _ptr(cache), _ptr(q), _ptr(qd), _ptr(qdd), _ptr(_f32(model["fixed_transforms"])),
            _ptr(_f32(model["link_masses_com"])), _ptr(_f32(model["link_inertias"])),
            _ptr(np.ascontiguousarray(model["joint_map_type"], np.int8)), _ptr(np.ascontiguousarray(model["joint_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_map"], np.int16)), _ptr(_f32(model["joint_offset_map"])),
            _ptr(_f32(gravity)), _ptr(level_links), None if fe is None else _ptr(fe), C.c_int(b), C.c_int(L), C.c_int(dof))
        return tau, cache

    def rnea_backward(self, grad_tau, q, qd, cache, model, gravity=(0, 0, 0, 0, 0, 9.81), want_f_ext_grad=False):
        q, qd, gt = _f32(q), _f32(qd), _f32(grad_tau)
        b, dof = q.shape
        L = model["fixed_transforms"].shape[0]
        g = [np.zeros((b, dof), np.float32) for _ in range(3)]
        gf = np.zeros((b, L, 6), np.float32) if want_f_ext_grad else None
        _, level_links = self.tree_levels(model["link_map"])
        self.lib.orc_rnea_backward(
            _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), None if gf is None else _ptr(gf), _ptr(gt), _ptr(q), _ptr(qd),
            _ptr(_f32(model["fixed_transforms"])), _ptr(_f32(model["link_masses_com"])), _ptr(_f32(model["link_inertias"])),
            _ptr(np.ascontiguousarray(model["joint_map_type"], np.int8)), _ptr(np.ascontiguousarray(model["joint_map"], np.int16)),
            _ptr(np.ascontiguousarray(model["link_map"], np.int16)), _ptr(_f32(model["joint_offset_map"])),
            _ptr(_f32(gravity)), _ptr(level_links), _ptr(_f32(cache)), C.c_int(b), C.c_int(L), C.c_int(dof))
        return (g[0], g[1], g[2], gf) if want_f_ext_grad else (g[0], g[1], g[2])

    # ------------------------------------------------------------------ Levenberg-Marquardt
    def lm_step(self, jacobian, jTerror, lambda_damping, joint_position_in):
        """(joint_position_out[b, d], pred_reduction[b]) of the reference's LevenbergMarquardtStep."""
        J = _f32(jacobian)
        b, r, d = J.shape
        q_out = np.zeros((b, d), np.float32)
        pred = np.zeros((b,), np.float32)
        self.lib.orc_lm_step(_ptr(q_out), _ptr(pred), _ptr(J), _ptr(_f32(jTerror)), _ptr(_f32(lambda_damping)),
                             _ptr(_f32(joint_position_in)), C.c_int(b), C.c_int(r), C.c_int(d))
        return q_out, pred

    # ------------------------------------------------------------------ optimiser step
    def lbfgs_step(self, step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0,
                   epsilon: float, stable_mode: bool):
        """In-place on (step_vec, rho/y/s buffers, x_0, grad_0); shapes as the reference."""
        m = y_buffer.shape[0]
        b = q.shape[0]
        v = int(np.prod(q.shape[1:]))
        for a in (step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        self.lib.orc_lbfgs_step(
            _ptr(step_vec), _ptr(rho_buffer), _ptr(y_buffer), _ptr(s_buffer), _ptr(q), _ptr(grad_q),
            _ptr(x_0), _ptr(grad_0), C.c_float(epsilon), C.c_int(b), C.c_int(m), C.c_int(v),
            C.c_int(int(stable_mode)),
        )
        return step_vec

    def line_search(self, state: Dict[str, np.ndarray], search_cost, search_action, search_gradient,
                    step_direction, search_magnitudes, c_1: float, c_2: float, strong_wolfe: bool,
                    approx_wolfe: bool, convergence_iteration: int, cost_delta_threshold: float,
                    cost_relative_threshold: float):
        """state keys (updated in place): best_cost, best_action, best_iteration,
        current_iteration, converged, exploration_cost/action/gradient, cost, action, gradient,
        exploration_idx, selected_idx."""
        b, nls = search_cost.shape[0], search_cost.shape[1]
        opt_dim = search_action.shape[-1]
        self.lib.orc_line_search(
            _ptr(state["best_cost"]), _ptr(state["best_action"]), _ptr(state["best_iteration"]),
            _ptr(state["current_iteration"]), _ptr(state["converged"]),
            C.c_int(convergence_iteration), C.c_float(cost_delta_threshold),
            C.c_float(cost_relative_threshold), _ptr(state["exploration_cost"]),
            _ptr(state["exploration_action"]), _ptr(state["exploration_gradient"]),
            _ptr(state["exploration_idx"]), _ptr(state["cost"]), _ptr(state["action"]),
            _ptr(state["gradient"]), _ptr(state["selected_idx"]), _ptr(_f32(search_cost)),
            _ptr(_f32(search_action)), _ptr(_f32(search_gradient)), _ptr(_f32(step_direction)),
            _ptr(_f32(search_magnitudes)), C.c_float(c_1), C.c_float(c_2),
            C.c_int(int(strong_wolfe)), C.c_int(int(approx_wolfe)), C.c_int(nls), C.c_int(opt_dim),
            C.c_int(b),
        )
        return state

    # ------------------------------------------------------------------ costs
    def tool_pose_distance(self, current_position, current_quat, goal_position, goal_quat, idxs_goal,
                           position_orientation_weight, terminal_axes_weight, non_terminal_axes_weight,
                           terminal_tolerance, non_terminal_tolerance, project_distance_to_goal,
                           rotation_method: int = 0):
        cp = _f32(current_position)
        b, h, L, _ = cp.shape
        gp = _f32(goal_position)
        ng = gp.shape[-2]
        out = {
            "distance": np.zeros((b, h, 2 * L), np.float32),
            "position_distance": np.zeros((b, h, L), np.float32),
            "rotation_distance": np.zeros((b, h, L), np.float32),
            "position_gradient": np.zeros((b, h, L, 3), np.float32),
            "rotation_gradient": np.zeros((b, h, L, 4), np.float32),
            "goalset_idx": np.zeros((b, h, L), np.int32),
        }
        self.lib.orc_tool_pose_distance(
            _ptr(out["distance"]), _ptr(out["position_distance"]), _ptr(out["rotation_distance"]),
            _ptr(out["position_gradient"]), _ptr(out["rotation_gradient"]), _ptr(out["goalset_idx"]),
            _ptr(cp), _ptr(_f32(current_quat)), _ptr(gp), _ptr(_f32(goal_quat)),
            _ptr(np.ascontiguousarray(idxs_goal, np.int32)), _ptr(_f32(position_orientation_weight)),
            _ptr(_f32(terminal_axes_weight)), _ptr(_f32(non_terminal_axes_weight)),
            _ptr(_f32(terminal_tolerance)), _ptr(_f32(non_terminal_tolerance)),
            _ptr(np.ascontiguousarray(project_distance_to_goal, np.uint8)),
            C.c_int(b), C.c_int(h), C.c_int(L), C.c_int(ng), C.c_int(rotation_method),
        )
        return out

    def cspace_position_cost(self, pos, p_b, weight, activation_distance, effort=None, effort_b=None,
                             cspace_target=None, cspace_target_idx=None, cspace_target_weight=0.0,
                             cspace_target_dof_weight=None, squared_l2_reg_weight=(0.0, 0.0),
                             current_position=None, current_velocity=None, idxs_current_state=None, v_b=None,
                             state_dt=None):
        p = _f32(pos)
        b, h, d = p.shape
        z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
        effort_b = _f32(effort_b) if effort_b is not None else np.stack([-1e9 * np.ones(d), 1e9 * np.ones(d)]).astype(np.float32)
        out_c, out_gp, out_gt = z(b, h, d), z(b, h, d), z(b, h, d)
        self.lib.orc_cspace_position_cost(
            _ptr(out_c), _ptr(out_gp), _ptr(out_gt), _ptr(p), _ptr(_f32(effort)) if effort is not None else None,
            _ptr(_f32(cspace_target) if cspace_target is not None else z(1, d)),
            _ptr(np.ascontiguousarray(cspace_target_idx if cspace_target_idx is not None else np.zeros(b), np.int32)),
            _ptr(_f32(p_b)), _ptr(effort_b), _ptr(_f32(weight)), _ptr(_f32(activation_distance)),
            _ptr(np.array([cspace_target_weight], np.float32)),
            _ptr(_f32(cspace_target_dof_weight) if cspace_target_dof_weight is not None else np.ones(d, np.float32)),
            _ptr(_f32(squared_l2_reg_weight)),
            _ptr(_f32(current_position) if current_position is not None else z(1, d)),
            _ptr(_f32(current_velocity) if current_velocity is not None else z(1, d)),
            _ptr(np.ascontiguousarray(idxs_current_state if idxs_current_state is not None else np.zeros(b), np.int32)),
            _ptr(_f32(v_b) if v_b is not None else z(2, d)),
            _ptr(_f32(state_dt) if state_dt is not None else z(1)),
            C.c_int(1), C.c_int(b), C.c_int(h), C.c_int(d),
        )
        return {"cost": out_c, "grad_position": out_gp, "grad_effort": out_gt}

    def cspace_state_cost(self, pos, vel, acc, jerk, state_dt, limits, weight, activation_distance, sql2_weights,
                          effort=None, target=None, idxs_target=None, target_weight=0.0, non_terminal_factor=1.0,
                          target_dof_weight=None, retime_weights=False, retime_regularization_weights=False):
        """reference CSpaceStateCost kernel; ``limits`` = dict position/velocity/acceleration/jerk/effort -> [2, dof]."""
        p = _f32(pos)
        b, h, d = p.shape
        z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
        big = np.stack([-1e9 * np.ones(d), 1e9 * np.ones(d)]).astype(np.float32)
        lim = [_f32(limits.get(k, big)) for k in ("position", "velocity", "acceleration", "jerk", "effort")]
        outs = [z(b, h, d) for _ in range(6)]
        self.lib.orc_cspace_state_cost(
            *[_ptr(o) for o in outs], _ptr(p), _ptr(_f32(vel)), _ptr(_f32(acc)), _ptr(_f32(jerk)),
            _ptr(_f32(effort)) if effort is not None else None, _ptr(_f32(state_dt)),
            _ptr(_f32(target) if target is not None else z(1, d)),
            _ptr(np.ascontiguousarray(idxs_target if idxs_target is not None else np.zeros(b), np.int32)),
            *[_ptr(x) for x in lim], _ptr(_f32(weight)), _ptr(_f32(activation_distance)), _ptr(_f32(sql2_weights)),
            _ptr(np.array([target_weight], np.float32)), _ptr(np.array([non_terminal_factor], np.float32)),
            _ptr(_f32(target_dof_weight) if target_dof_weight is not None else np.ones(d, np.float32)),
            C.c_int(1), C.c_int(b), C.c_int(h), C.c_int(d), C.c_int(int(retime_weights)),
            C.c_int(int(retime_regularization_weights)))
        keys = ("cost", "grad_position", "grad_velocity", "grad_acceleration", "grad_jerk", "grad_effort")
        return dict(zip(keys, outs))

    def trajectory_cost_sum(self, self_cost, scene_cost):
        sc = _f32(scene_cost)
        b, h, S = sc.shape
        out = np.zeros((b,), np.float32)
        self.lib.orc_trajectory_cost_sum(
            _ptr(out), _ptr(_f32(self_cost)) if self_cost is not None else None, _ptr(sc),
            C.c_int(b), C.c_int(h), C.c_int(S),
        )
        return out


_ORACLE: Optional[Oracle] = None


def load_native_oracle() -> Oracle:
    """bench.py's CPU-baseline leg: the -O3 -march=native build (see build_native_oracle)."""
    return Oracle(build_native_oracle())


def load_oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
