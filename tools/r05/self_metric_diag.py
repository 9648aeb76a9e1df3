"""self-collision of a held configuration: collision checker vs the trajectory optimiser's metrics rollout"""
import sys
import torch
from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
from curobo_amd.collision_checking import RobotCollisionChecker
from curobo_amd.types import JointState

robot = sys.argv[1] if len(sys.argv) > 1 else "dual_ur10e"
planner = MotionPlanner(MotionPlannerCfg.create(robot=f"{robot}.yml", scene_model="collision_table.yml"))
planner.warmup()
torch.manual_seed(3)
q = planner.sample_configs(8, rejection_ratio=50)[:4]
chk = RobotCollisionChecker(planner.config.trajopt_solver_config.kinematics, planner.config.trajopt_solver_config.scene)
dw, ds = chk.get_scene_self_collision_distance_from_joints(q)
print("checker: self", ds.view(-1).tolist(), "scene", dw.sum(-1).view(-1).tolist())
slv = planner.trajopt_solver._solver
for name, ro in (("metrics", slv.metrics_rollout), ("optimiser", slv.rollout)):
    B, nk, D = ro.batch_size, ro.cfg.n_knots, q.shape[1]
    for i in range(2):
        start = q[i:i + 1]
        ro.update_start_state(start.expand(slv.P, D).contiguous(), start_idx=None) if False else None
    print(name, "batch", B, "rows per call", B)
m = slv.metrics_rollout
D = q.shape[1]
for i in range(4):
    cur = JointState.from_position(q[i:i + 1].clone(), planner.joint_names)
    goal = planner.compute_kinematics(cur).tool_poses.as_goal()  # stay where you are
    r = planner.plan_pose(goal, cur, max_attempts=1)
    H = m.cfg.padded_horizon
    sd = m.self_dist.view(-1, H)
    print(i, "plan to own pose: success", None if r is None else r.success.view(-1).tolist(), "self_dist per seed (sum over horizon)", sd.sum(-1).tolist(),
          "first points", sd[0, :4].tolist(), "max joint travel", float((m.position[0] - q[i]).abs().max()))
