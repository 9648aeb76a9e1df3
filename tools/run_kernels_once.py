"""Every kernel the BASELINE configs launch, a few plain launches each on one stream (no hipGraph, no side streams): the
target of the rocprofv3 counter passes of tools/collect_profiles_r03.sh.  Workloads and shapes are bench.py's:

  c2   Franka, 256 seeds x 4 candidates x 33 points, 4-cuboid world: the drop-in kernel sequence, the fused launch
       (1024 and 256 trajectories), the optimiser's iteration tail (workgroup and wavefront form)
  c3   UR10e, 512 x 4 x 33, 128^3 fp16 ESDF: FK, self, scene_collision_packed_kernel, FK VJP, the fused launch
  c4   Unitree G1, 256 x 4 x 33: FK, self_collision_tiles2_kernel, RNEA forward (staged kernels) / backward, c-space cost, FK VJP
  c5   Franka, 2 worlds (cuboids + 64^3 ESDF) x 512 x 4 x 65: the fused multi-env launch, the swept scene kernel

  mesh bench.py's mesh world (the C2 cuboids as 12 288 triangles): sphere_mesh_select_kernel + sphere_mesh_cells_kernel (+ the
       tree walk of what the cell lists cannot answer), and the same launch over meshes without lists (sphere_mesh_walk_kernel)
  ik   C1 (the IK half of the metric): Franka, 100 problems x 64 seeds, 4-cuboid world: seed_ik_solve_kernel (128 Levenberg-
       Marquardt runs per problem, 16 iterations), ik_rank_kernel, and the IK rollout rollout_ik_fused_kernel of one L-BFGS
       iteration (100 x 64 x 4 line-search candidates)

Usage: python tools/run_kernels_once.py [c2] [c3] [c4] [c5] [mesh] [trajopt] [ik] [--reps N]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B_  # noqa: E402  (the stage helpers of bench.py: rollout_self / rollout_scene / rollout_bwd_fk / _fk_fwd / _fk_bwd)

dev = torch.device("cuda:0")
REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5


def run(fn, reps=None):
    for _ in range(reps or REPS):
        fn()
    torch.cuda.synchronize()


def collision_rollout(robot, scene_arrays, batch, **cfg_kw):
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import start_configuration

    kcfg = KinematicsCfg.from_packaged(robot, device=dev)
    model, kin = kcfg.model, kcfg.kinematics_config
    scene = SceneData.from_arrays(scene_arrays, dev) if scene_arrays is not None else None
    out = []
    for fused in (False, True):
        ro = CollisionRollout(kin, scene, batch, CollisionRolloutCfg(use_fused=fused, **cfg_kw))
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
        out.append(ro)
    return model, kin, out[0], out[1]


def sequence_and_fused(model, seq, fused, seed, env_idx=None):
    from curobo_amd.workloads import seed_knots

    B = seq.batch_size
    x = torch.as_tensor(seed_knots(model, B, seq.cfg.n_knots, seed=seed), device=dev).reshape(B, -1)
    act = x.view(B, seq.cfg.n_knots, -1)
    for ro in (seq, fused):
        ro.update_env_query_idx(env_idx)
    seq.evaluate_action(act)
    seq.backward()
    torch.cuda.synchronize()
    run(lambda: seq.compute_state_from_action(act))
    run(lambda: seq.compute_kinematics(seq.position))
    run(lambda: B_.rollout_self(seq))
    run(lambda: B_.rollout_scene(seq))
    run(lambda: B_.rollout_bwd_fk(seq))
    run(seq.backward)  # (FK VJP + B-spline VJP)
    if fused.fused_available():
        run(lambda: fused.cost_and_gradient(x))
    return x


def c2():
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
    from curobo_amd.rollout import CollisionRollout
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    model, kin, seq, fused = collision_rollout("franka", cuboid_scene_arrays(c2_world()), 1024)
    sequence_and_fused(model, seq, fused, 2)
    small = CollisionRollout(kin, fused.scene, 256, fused.cfg)  # one seed shard of the default command
    small.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    xs = torch.as_tensor(seed_knots(model, 256, 12, seed=2), device=dev).reshape(256, -1)
    run(lambda: small.cost_and_gradient(xs))
    bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
    for problems, ro, overlapped in ((256, fused, False), (64, small, False), (64, small, True)):
        opt = LBFGSOpt(LBFGSOptCfg(num_problems=problems), ro.cost_and_gradient, 12, kin.num_dof, bounds, dev, use_cuda_graph=False)
        opt.overlapped = overlapped
        opt.reinitialize(torch.as_tensor(seed_knots(model, problems, 12, seed=2), device=dev))
        run(opt._opt_step)


def c3():
    from curobo_amd.workloads import c3_voxel_world

    model, kin, seq, fused = collision_rollout("ur10e", c3_voxel_world(), 2048)
    sequence_and_fused(model, seq, fused, 4)


def c4():
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.workloads import seed_knots, start_configuration

    kcfg = KinematicsCfg.from_packaged("unitree_g1", device=dev)
    model, kin = kcfg.model, kcfg.kinematics_config
    B, H = 1024, 33
    D, S = kin.num_dof, kin.num_spheres
    ro = TrajOptRollout(kin, None, B, TrajOptRolloutCfg(use_fused=False, use_torque_limits=True, effort_limit=[200.0] * D))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    x = torch.as_tensor(seed_knots(model, B, 12, seed=6, spread=0.15), device=dev).reshape(B, -1)
    run(lambda: ro.cost_and_gradient(x), 3)  # the whole launch set (every kernel of the C4 rollout)


def c5():
    from curobo_amd.workloads import c5_mixed_worlds

    n_prob, seeds, nls = 2, 512, 4
    B = n_prob * seeds * nls
    env = torch.arange(n_prob, dtype=torch.int32, device=dev).repeat_interleave(seeds * nls)
    model, kin, seq, fused = collision_rollout("franka", c5_mixed_worlds(n_prob, voxels=True), B, interpolation_steps=4)
    sequence_and_fused(model, seq, fused, 8, env)


def mesh():
    """bench.py's mesh world: the C2 cuboids as four triangle meshes (12 288 triangles), C2 shapes, swept + speed metric"""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import seed_knots, start_configuration

    kcfg = KinematicsCfg.from_packaged("franka", device=dev)
    model, kin = kcfg.model, kcfg.kinematics_config
    scene = SceneData.from_arrays(None, dev, meshes=B_.c2_world_as_meshes())
    B = 1024
    cfg = CollisionRolloutCfg(use_fused=False)
    ro = CollisionRollout(kin, scene, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
    ro.compute_kinematics(ro.compute_state_from_action(x.view(B, cfg.n_knots, -1)))
    run(lambda: Cn.sphere_obstacle_collision(ro.scene_dist, ro.scene_grad, ro.robot_spheres, scene.struct, ro._w_scene, ro._eta,
                                             ro.env_query_idx, B, cfg.padded_horizon, kin.num_spheres, False, 3, True, ro._speed_dt))
    from curobo_amd.scene import MeshStore

    walk = SceneData.from_arrays(None, dev, meshes=MeshStore(B_.c2_world_as_meshes(), dev, cells=False))
    run(lambda: Cn.sphere_obstacle_collision(ro.scene_dist, ro.scene_grad, ro.robot_spheres, walk.struct, ro._w_scene, ro._eta,
                                             ro.env_query_idx, B, cfg.padded_horizon, kin.num_spheres, False, 3, True, ro._speed_dt))


def ik():
    """C1 shapes: the Levenberg-Marquardt seed solver (seed_ik_solve_kernel + ik_rank_kernel) as bench.py's `ik` object times it,
    and one evaluation of the IK rollout over 100 x 64 x 4 rows (rollout_ik_fused_kernel), plain launches"""
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import IKSolver, IKSolverCfg
    from curobo_amd.workloads import c1_world, feasible_goals

    kcfg = KinematicsCfg.from_packaged("franka", device=dev)
    kin = kcfg.kinematics_config
    scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
    P, S = 100, 64
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, stream_shards=1, use_cuda_graph=False)
                      if "use_cuda_graph" in IKSolverCfg.__dataclass_fields__ else IKSolverCfg(num_seeds=S, stream_shards=1))
    gp, gq = feasible_goals(kin, scene, P)
    T = kin.num_pose_links
    gp4, gq4 = gp.to(dev).view(P, T, 1, 3).contiguous(), gq.to(dev).view(P, T, 1, 4).contiguous()
    ss = solver.seed_solver
    run(lambda: ss.solve_batch(gp4, gq4, return_seeds=S))
    run(lambda: solver.solve_pose(gp, gq, exit_early=False), 2)  # (every kernel of a full solve, the IK rollout among them)


def trajopt():
    """the fused launch with the FULL trajectory-optimisation cost set (tool pose + c-space STATE + self + swept scene), C2 shapes:
    1024 trajectories (the bench's `full_trajopt_rollout`) and 32 (one problem x 8 seeds x 4 candidates: a planner's iteration)"""
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    kcfg = KinematicsCfg.from_packaged("franka", device=dev)
    model, kin = kcfg.model, kcfg.kinematics_config
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
    for B in (1024, 32):
        ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg())
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
        x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=dev).reshape(B, -1)
        run(lambda: ro.cost_and_gradient(x))


if __name__ == "__main__":
    known = ("c2", "c3", "c4", "c5", "mesh", "trajopt", "ik")
    want = [a for a in sys.argv[1:] if a in known] or list(known)
    for w in want:
        globals()[w]()
    print("ran", want)
