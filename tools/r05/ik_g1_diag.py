"""why do G1 IK problems fail: errors of the failures, the LM stage alone, more seeds"""
import sys
import numpy as np
import torch
from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
from curobo_amd.types import JointState

robot = sys.argv[1] if len(sys.argv) > 1 else "unitree_g1"
for seeds, iters in ((2, 240), (8, 240), (16, 240), (2, 500)):
    cfg = InverseKinematicsCfg.create(robot=f"{robot}.yml", scene_model=None, num_seeds=seeds, position_tolerance=0.005, self_collision_check=False,
                                      use_cuda_graph=True, seed_solver_num_seeds=128, max_batch_size=100, override_iters_for_multi_link_ik=iters)
    ik = InverseKinematics(cfg)
    torch.manual_seed(2)
    q = ik.sample_configs(100, rejection_ratio=50)[:100].contiguous()
    g = ik.compute_kinematics(JointState.from_position(q)).tool_poses.as_goal()
    ik.config.exit_early = False
    r = ik.solve_pose(g)
    ok = r.success.view(-1)
    pe, re = r.position_error.view(-1).cpu().numpy(), r.rotation_error.view(-1).cpu().numpy()
    bad = ~ok.cpu().numpy()
    slv = ik._solver(100)
    print(f"seeds {seeds} iters {iters}: success {float(ok.float().mean()):.3f}; failures: pos err {np.sort(pe[bad])[-8:].round(4)} rot err {np.sort(re[bad])[-8:].round(4)}", flush=True)
    print("   failures with pos < 5 mm and rot < 0.05:", int(((pe < 0.005) & (re < 0.05) & bad).sum()), "of", int(bad.sum()))
    # LM stage alone
    ss = slv.seed_solver
    if ss is not None:
        gp, gq = g.static_goals(ik.tool_frames)
        res = ss.solve_batch(gp.reshape(100, -1, 1, 3).contiguous(), gq.reshape(100, -1, 1, 4).contiguous(), return_seeds=1) if hasattr(ss, "solve_batch") else None
        if res is not None:
            print("   LM stage alone: success", float(res.success.float().mean()), "seeds", ss.S, "iterations", ss.cfg.max_iterations)
    # feasibility of the solutions: limits
    lim = ik.kinematics.kinematics_config.joint_limits_position
    sol = r.solution.view(100, -1)
    print("   solutions outside limits:", int(((sol < lim[0] - 1e-4) | (sol > lim[1] + 1e-4)).any(-1).sum()))
