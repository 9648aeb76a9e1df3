"""Closed-loop MPC on the scenario of tests/test_gpu_mpc.py::test_mpc_tracks_a_pose_goal_in_closed_loop (two Franka arms, C1 world, pose
goals = FK of configurations near the start, 400 command steps with perfect tracking) under this package's MPC task values and under the
reference's (content/configs/task/mpc/lbfgs_mpc.yml:5-52 = MPCSolverCfg.reference_task()): final / halfway position error, steps to 5 mm,
solve times, feasibility, collisions of the executed states (oracle).   python tools/r06/mpc_task_compare.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.kinematics import Kinematics, KinematicsCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.solver import MPCSolver, MPCSolverCfg  # noqa: E402
from curobo_amd.types import JointState  # noqa: E402
from curobo_amd.workloads import c1_world, start_configuration  # noqa: E402
from oracle import load_oracle  # noqa: E402  (checker only)

dev = torch.device("cuda:0")
kcfg = KinematicsCfg.from_packaged("franka", device=dev)
kin = kcfg.kinematics_config
arrays = cuboid_scene_arrays(c1_world())
scene = SceneData.from_arrays(arrays, dev)
fk = Kinematics(kcfg, compute_spheres=True)
orc = load_oracle()
B = 2
q0 = torch.as_tensor(start_configuration(kcfg.model), device=dev).repeat(B, 1)
dq = torch.tensor([[0.5, 0.2, -0.3, 0.3, 0.2, -0.2, 0.3], [-0.5, 0.1, 0.3, 0.2, -0.3, 0.3, -0.2]], device=dev)
goal = fk.compute_kinematics(JointState.from_position((q0 + dq).unsqueeze(1))).tool_poses.as_goal()
out = {}
for label, make in (("package_defaults", lambda c: MPCSolverCfg(continuous_commands=c)), ("reference_task", lambda c: MPCSolverCfg.reference_task(continuous_commands=c))):
    for continuous in (False, True):
        mpc = MPCSolver(kin, scene, B, make(continuous))
        state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0))
        mpc.setup(state, goal)
        errs, times, feas, coll = [], [], 0, 0.0
        for step in range(400):
            res = mpc.optimize_next_action(state)
            a = res.next_action
            state = JointState(position=a.position, velocity=a.velocity, acceleration=a.acceleration)
            st = fk.compute_kinematics(JointState.from_position(state.position.unsqueeze(1)))
            errs.append(float((st.tool_poses.position[:, 0, 0] - goal.position[:, 0, 0, 0]).norm(dim=-1).max()))
            feas += int(bool(res.feasible.all()))
            if res.reoptimized:
                times.append(float(res.solve_time))
            if step % 20 == 19:
                coll += float(orc.scene_collision(st.robot_spheres.cpu().numpy(), arrays, 1.0, 0.0)["distance"].sum())
        e = np.asarray(errs)
        to5 = int(np.argmax(e < 0.005)) if (e < 0.005).any() else None
        out[f"{label}{'_continuous' if continuous else ''}"] = {
            "error_m_start": round(errs[0], 4), "error_m_step_200": round(errs[199], 5), "error_m_final": round(errs[-1], 6),
            "first_step_within_5mm": to5, "feasible_steps": feas, "collision_cost_of_sampled_states": coll,
            "cold_solve_ms": round(1e3 * times[0], 2), "warm_solve_ms_median": round(1e3 * float(np.median(times[1:])), 2)}
        print(label, continuous, out[f"{label}{'_continuous' if continuous else ''}"], flush=True)
# ---- the front end as the reference is used (update_goal_tool_poses runs the goal's IK and tracks the solution in joint space too)
from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg  # noqa: E402
from curobo_amd.scene.types import Cuboid, SceneCfg  # noqa: E402

world = SceneCfg(cuboid=[Cuboid(name=f"c{i}", dims=list(o["dims"]), pose=list(o["pose"])) for i, o in enumerate(c1_world()[0])])
for task in ("package", "reference"):
    mpc = ModelPredictiveControl(ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=world, max_batch_size=B, task=task))
    state = JointState(position=q0.clone(), velocity=torch.zeros_like(q0), acceleration=torch.zeros_like(q0), joint_names=mpc.joint_names)
    mpc.setup(state)
    ok = mpc.update_goal_tool_poses(goal)
    errs, times = [], []
    for step in range(400):
        res = mpc.optimize_next_action(state)
        a = res.next_action
        state = JointState(position=a.position.clone(), velocity=a.velocity.clone(), acceleration=a.acceleration.clone(), joint_names=mpc.joint_names)
        st = fk.compute_kinematics(JointState.from_position(state.position.unsqueeze(1)))
        errs.append(float((st.tool_poses.position[:, 0, 0] - goal.position[:, 0, 0, 0]).norm(dim=-1).max()))
        if res.reoptimized:
            times.append(float(res.solve_time))
    e = np.asarray(errs)
    out[f"front_end_with_goal_ik_{task}"] = {
        "goal_ik_ok": bool(ok), "error_m_step_200": round(errs[199], 5), "error_m_final": round(errs[-1], 6),
        "first_step_within_5mm": int(np.argmax(e < 0.005)) if (e < 0.005).any() else None,
        "cold_solve_ms": round(1e3 * times[0], 2), "warm_solve_ms_median": round(1e3 * float(np.median(times[1:])), 2)}
    print("front end", task, out[f"front_end_with_goal_ik_{task}"], flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
