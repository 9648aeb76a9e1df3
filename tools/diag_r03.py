"""GPU diagnostics of round 3 (not a test): (1) which check fails the trajopt seeds when every seed shares one IK goal,
(2) where scene_collision_packed_kernel and the oracle differ per sphere on the C3 world."""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402
from oracle import load_oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = load_oracle()


def trajopt_k1():
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import c2_world, start_configuration

    model = load_model("franka"); kin = KinematicsParams.from_model(model, dev)
    arrays = cuboid_scene_arrays(c2_world()); scene = SceneData.from_arrays(arrays, dev)
    md = model.as_dict(); P = 6
    cand = sample_q(model, 400, seed=12, scale=0.6)
    fk = oracle.kinematics_forward(cand, md)
    sph = fk["robot_spheres"].reshape(400, 1, -1, 4)
    free = (oracle.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0) & \
        (oracle.scene_collision(sph, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0)
    sel = np.nonzero(free)[0][:P]
    gp, gq = fk["link_pos"][sel, 0], fk["link_quat"][sel, 0]
    start = start_configuration(model)
    for K in (1, 4):
        for fa, chk in ((1, True),):
            cfg = TrajOptSolverCfg(num_seeds=4, num_ik_goals=K, check_interpolated=chk)
            slv = TrajOptSolver(kin, scene, P, cfg)
            r = slv.solve_pose(torch.as_tensor(start), torch.as_tensor(gp), torch.as_tensor(gq), finetune_attempts=fa)
            a = r.all_seeds
            print(f"K={K} finetune={fa} interp_check={chk}: success {r.success.float().mean():.2f} passes {r.finetune_passes}")
            for k in ("success", "in_limits", "no_self_collision", "no_scene_collision", "feasible_interpolated", "converged"):
                print("   ", k, a[k].int().tolist())
            for t in slv.last_pass_trace:
                for k, v in t.items():
                    print("      ", k, np.round(v.float().cpu().numpy(), 3).tolist())


def c3_per_sphere():
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c3_voxel_world, seed_knots, start_configuration

    model = load_model("ur10e"); kin = KinematicsParams.from_model(model, dev)
    arrays = c3_voxel_world()
    B = 256
    x = torch.as_tensor(seed_knots(model, B, 12, seed=4), device=dev).reshape(B, -1)
    for coarse in (True, False):
        scene = SceneData.from_arrays(arrays, dev, coarse_culling=coarse)
        for sweep, speed in ((True, True), (True, False), (False, False)):
            seq = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_fused=False, use_sweep=sweep, use_speed_metric=speed))
            seq.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
            seq.compute_kinematics(seq.compute_state_from_action(x.view(B, 12, -1)))
            seq.compute_costs(); torch.cuda.synchronize()
            sph = seq.robot_spheres.cpu().numpy(); d = seq.scene_dist.cpu().numpy()
            c = seq.cfg
            wc = oracle.scene_collision(sph, arrays, c.scene_collision_weight, c.activation_distance, sweep=sweep,
                                        enable_speed_metric=speed, speed_dt=c.traj_dt)
            ref = wc["distance"]
            err = np.abs(d - ref); rel = err / np.maximum(np.abs(ref), 1.0)
            bad = rel > 1e-5
            print(f"coarse={coarse} sweep={sweep} speed={speed}: max abs {err.max():.4f} max rel {rel.max():.2e} bad {bad.sum()} of {(ref > 0).sum()} hits")
            if bad.any():
                idx = np.argwhere(bad)
                o = np.argsort(-rel[bad])[:6]
                for b, h, s in idx[o]:
                    p = sph[b, max(h - 1, 0):h + 2, s]
                    print("     b,h,s", b, h, s, "hip", d[b, h, s], "ref", ref[b, h, s], "r", sph[b, h, s, 3], "step", np.linalg.norm(np.diff(p[:, :3], axis=0), axis=-1))


if __name__ == "__main__":
    what = sys.argv[1:] or ["trajopt", "c3"]
    if "c3" in what:
        c3_per_sphere()
    if "trajopt" in what:
        trajopt_k1()
