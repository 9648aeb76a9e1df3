"""solver-level scenarios that no test runs yet, one after the other (each in its own try: a Python error is reported and the
sweep goes on; a device fault ends the process and names the scenario last printed)"""
import os
import sys
import time
import traceback

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
from test_oracle_mesh import box_shape, sphere_shape  # noqa: E402

from curobo_amd.motion_planner import BatchMotionPlanner, MotionPlanner, MotionPlannerCfg  # noqa: E402
from curobo_amd.scene import cuboid_scene_arrays  # noqa: E402
from curobo_amd.types import JointState  # noqa: E402
from oracle.oracle import Oracle, mesh_scene_arrays  # noqa: E402

orc = Oracle()
dev = "cuda"
vb, fb = box_shape([0.16, 0.16, 0.7], 2)
vs, fs = sphere_shape(0.12)
table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
pillar_pose = [0.5, 0.0, 0.35, 1, 0, 0, 0]
Q0 = [-0.9, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]
only = set(filter(None, os.environ.get("ONLY", "").split(",")))


def free(model, traj, arrays, env=None):
    n, H, D = traj.shape
    sph = orc.kinematics_forward(traj.reshape(n * H, D), model.as_dict(), horizon=H)["robot_spheres"].reshape(n, H, -1, 4)
    kw = dict(env_query_idx=np.asarray(env, np.int32), use_multi_env=True) if env is not None else {}
    return orc.scene_collision(sph, arrays, 1.0, 0.0, **kw)["distance"].sum((1, 2))


def scenario(name):
    def deco(fn):
        if only and name not in only:
            return fn
        print(f"== {name}", flush=True)
        t0 = time.time()
        try:
            fn()
            torch.cuda.synchronize()
            print(f"   ok ({time.time() - t0:.1f} s)", flush=True)
        except Exception:  # noqa: BLE001
            traceback.print_exc()
            print("   FAILED", flush=True)
        return fn
    return deco


def mesh_world(pose=pillar_pose):
    return {"cuboid": {"table": table}, "mesh": {"pillar": {"vertices": vb, "faces": fb, "pose": pose}}}


def start_goal(planner, b=1, spread=0.0):
    q0 = torch.tensor([Q0], device=dev).repeat(b, 1)
    if b > 1:
        q0[:, 1] += torch.linspace(-spread, spread, b, device=dev)
    cur = JointState.from_position(q0, planner.joint_names)
    g = cur.clone()
    g.position[:, 0] = 0.9
    return cur, g, planner.compute_kinematics(g).tool_poses.as_goal()


@scenario("mesh pose update under captured graphs")
def _():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=mesh_world(), num_ik_seeds=32, num_trajopt_seeds=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    scene = config.trajopt_solver_config.scene
    cur, gjs, goal = start_goal(planner)
    out = []
    for where in (pillar_pose, [0.5, 0.0, 3.35, 1, 0, 0, 0], pillar_pose):
        scene.update_obstacle_pose("pillar", where)
        arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[{"name": "pillar", "vertices": vb, "faces": fb, "pose": where}]])}
        res = planner.plan_pose(goal, cur, max_attempts=4)
        ok = res is not None and bool(res.success[0, 0])
        d = float(free(model, res.js_solution.position[0].cpu().numpy(), arrays)[0]) if res is not None else None
        out.append((ok, None if res is None else round(float(res.motion_time[0, 0]), 3), d))
    print("   (success, motion time, oracle collision) pillar in place / lifted away / back:", out)
    assert all(o[0] and o[2] == 0.0 for o in out), out
    assert out[1][1] < out[0][1] and out[1][1] < out[2][1], "without the pillar the motion is shorter"


@scenario("update_world to a different mesh scene")
def _():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=mesh_world(), num_ik_seeds=32, num_trajopt_seeds=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    cur, gjs, goal = start_goal(planner)
    r1 = planner.plan_pose(goal, cur, max_attempts=4)
    w2 = {"cuboid": {"table": table}, "mesh": {"ball": {"vertices": vs, "faces": fs, "pose": [0.55, 0.0, 0.4, 1, 0, 0, 0]}}}
    planner.update_world(w2)
    r2 = planner.plan_pose(goal, cur, max_attempts=4)
    arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[{"name": "ball", "vertices": vs, "faces": fs, "pose": [0.55, 0.0, 0.4, 1, 0, 0, 0]}]])}
    ok = [r is not None and bool(r.success[0, 0]) for r in (r1, r2)]
    print("   success", ok, "second plan against the NEW world:", float(free(model, r2.js_solution.position[0].cpu().numpy(), arrays)[0]))
    assert all(ok) and float(free(model, r2.js_solution.position[0].cpu().numpy(), arrays)[0]) == 0.0


@scenario("batch planner, one mesh world per problem")
def _():
    B = 4
    poses = [[0.5, 0.0, 0.35, 1, 0, 0, 0], [0.5, 0.1, 0.35, 1, 0, 0, 0], [0.45, -0.1, 0.35, 1, 0, 0, 0], [0.5, 0.0, 3.0, 1, 0, 0, 0]]
    worlds = [mesh_world(p) for p in poses]
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=worlds, max_batch_size=B, multi_env=True, num_ik_seeds=32, num_trajopt_seeds=4)
    planner = BatchMotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    cur, gjs, goal = start_goal(planner, B, 0.1)
    res = planner.plan_pose(goal, cur, max_attempts=4)
    arrays = {**cuboid_scene_arrays([[table]] * B), **mesh_scene_arrays([[{"name": "pillar", "vertices": vb, "faces": fb, "pose": p}] for p in poses])}
    succ = res.success[:, 0].cpu().numpy()
    d = free(model, res.js_solution.position[:, 0].cpu().numpy(), arrays, env=np.arange(B))
    print("   success", succ.tolist(), "oracle collision per problem in ITS world", d.tolist(), "motion time", res.motion_time[:, 0].cpu().numpy().round(3).tolist())
    assert succ.sum() >= 3 and (d[succ] == 0).all()


@scenario("goal set in a mesh world")
def _():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=mesh_world(), num_ik_seeds=32, num_trajopt_seeds=4, max_goalset=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    cur, gjs, goal = start_goal(planner)
    from curobo_amd.types import GoalToolPose
    # four candidates: the goal, and three shifted copies (one inside the pillar: unreachable without collision)
    pos = goal.position.repeat(1, 1, 1, 4, 1).clone()
    quat = goal.quaternion.repeat(1, 1, 1, 4, 1).clone()
    pos[0, 0, 0, 1] = torch.tensor([0.5, 0.0, 0.3], device=dev)
    pos[0, 0, 0, 2, 2] += 0.1
    pos[0, 0, 0, 3, 1] -= 0.1
    gs = GoalToolPose(goal.tool_frames, pos, quat)
    res = planner.plan_pose(gs, cur, max_attempts=4)
    arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[{"name": "pillar", "vertices": vb, "faces": fb, "pose": pillar_pose}]])}
    ok = res is not None and bool(res.success[0, 0])
    print("   success", ok, "goalset index", None if res is None else res.goalset_index.view(-1).tolist() if getattr(res, "goalset_index", None) is not None else "n/a")
    assert ok and float(free(model, res.js_solution.position[0].cpu().numpy(), arrays)[0]) == 0.0


@scenario("plan_grasp in a mesh world")
def _():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=mesh_world(), num_ik_seeds=32, num_trajopt_seeds=4, max_goalset=2)
    planner = MotionPlanner(config)
    cur, gjs, goal = start_goal(planner)
    from curobo_amd.types import GoalToolPose
    pos = goal.position.repeat(1, 1, 1, 2, 1).clone()
    quat = goal.quaternion.repeat(1, 1, 1, 2, 1).clone()
    pos[0, 0, 0, 1, 2] += 0.05
    r = planner.plan_grasp(GoalToolPose(goal.tool_frames, pos, quat), cur)
    print("   status", r.status, "success", r.success.view(-1).tolist(), "approach / grasp / lift", r.approach_success.view(-1).tolist(),
          r.grasp_success.view(-1).tolist(), r.lift_success.view(-1).tolist())
    assert bool(r.success.any())


@scenario("no graphs, mesh world")
def _():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=mesh_world(), num_ik_seeds=32, num_trajopt_seeds=4, use_cuda_graph=False)
    planner = MotionPlanner(config)
    cur, gjs, goal = start_goal(planner)
    res = planner.plan_pose(goal, cur, max_attempts=4)
    assert res is not None and bool(res.success[0, 0])


@scenario("dual arm in a mesh world")
def _():
    w = {"cuboid": {"table": table}, "mesh": {"ball": {"vertices": vs, "faces": fs, "pose": [0.6, 0.0, 0.5, 1, 0, 0, 0]}}}
    config = MotionPlannerCfg.create(robot="dual_ur10e.yml", scene_model=w, num_ik_seeds=32, num_trajopt_seeds=4)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    q0 = planner.default_joint_state.position.view(1, -1).clone()
    cur = JointState.from_position(q0, planner.joint_names)
    g = cur.clone()
    g.position[0, 0] += 0.5
    g.position[0, 6] -= 0.5
    goal = planner.compute_kinematics(g).tool_poses.as_goal()
    res = planner.plan_pose(goal, cur, max_attempts=4)
    ok = res is not None and bool(res.success[0, 0])
    arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[{"name": "ball", "vertices": vs, "faces": fs, "pose": [0.6, 0.0, 0.5, 1, 0, 0, 0]}]])}
    print("   success", ok, "oracle collision", None if res is None else float(free(model, res.js_solution.position[0].cpu().numpy(), arrays)[0]))
    assert ok


@scenario("mixed scenes per problem")
def _():
    from curobo_amd.scene import voxel_grid_from_sdf
    from curobo_amd.scene.config import voxel_arrays_from_config

    B = 4
    centre, half = np.array([0.5, 0.0, 0.35]), np.array([0.08, 0.08, 0.35])

    def pillar(p):
        q = np.abs(p - centre) - half
        return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)

    gpose = [0.5, 0.0, 0.45, 1, 0, 0, 0]
    grid = voxel_grid_from_sdf(pillar, (32, 32, 48), 0.02, pose7=gpose, max_distance=10.0)
    vox = {"pillar": {"dims": [0.64, 0.64, 0.96], "voxel_size": 0.02, "pose": gpose, "feature_tensor": grid["voxel_features"].reshape(-1)}}
    ball = {"ball": {"vertices": vs, "faces": fs, "pose": [0.0, 0.55, 0.9, 1, 0, 0, 0]}}
    worlds = []
    for i in range(B):  # pillar as a voxel grid (0, 2) or as a mesh (1, 3); a ball mesh above in every world
        w = {"cuboid": {"table": table}, "mesh": dict(ball)}
        if i % 2 == 0:
            w["voxel"] = vox
        else:
            w["mesh"]["pillar"] = {"vertices": vb, "faces": fb, "pose": pillar_pose}
        worlds.append(w)
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=worlds, max_batch_size=B, multi_env=True, num_ik_seeds=32, num_trajopt_seeds=4)
    planner = BatchMotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    cur, gjs, goal = start_goal(planner, B, 0.1)
    res = planner.plan_pose(goal, cur, max_attempts=4)
    succ = res.success[:, 0].cpu().numpy()
    # oracle: every world on its own (the kinds differ per world)
    d = []
    for i in range(B):
        arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[dict(m, name=k) for k, m in worlds[i]["mesh"].items()]])}
        if "voxel" in worlds[i]:
            arrays.update(voxel_arrays_from_config(worlds[i]))
        d.append(float(free(model, res.js_solution.position[i:i + 1, 0].cpu().numpy(), arrays)[0]))
    print("   success", succ.tolist(), "oracle collision per problem in ITS world", d, "motion time", res.motion_time[:, 0].cpu().numpy().round(3).tolist())
    assert succ.sum() >= 3 and all(x == 0.0 for x, ok in zip(d, succ) if ok)
