"""Per-phase timing of the fused rollout kernel on the C2 workload (development tool).

Uses the library's profile hook (curobo_hip_rollout_fused_set_profile_buffer): every workgroup
stamps the 100 MHz wall clock at its phase boundaries.  Prints the mean/median duration of each
phase per workgroup plus the launch duration measured with HIP events.
    python tools/profile_fused.py [--batch 1024]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--trajopt", action="store_true",
                    help="full trajopt cost set (pose + c-space STATE terms); needs a library built with "
                         "CUROBO_HIP_EXTRA_FLAGS=-DCUROBO_FUSED_STAMP_TERMS")
    args = ap.parse_args()
    from curobo_amd._lib import load
    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    dev = torch.device("cuda:0")
    model = load_packaged_robot("franka")
    kin = KinematicsParams.from_model(model, dev)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
    cfg = CollisionRolloutCfg(use_sweep=not args.no_sweep, use_speed_metric=not args.no_sweep)
    B = args.batch
    if args.trajopt:
        from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
        cfg = TrajOptRolloutCfg(use_fused=True)
        ro = TrajOptRollout(kin, scene, B, cfg)
    else:
        ro = CollisionRollout(kin, scene, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
    for _ in range(3):
        ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ro.cost_and_gradient(x)
    e1.record()
    torch.cuda.synchronize()
    print(f"fused launch: {e0.elapsed_time(e1) * 1e3 / args.reps:.1f} us for {B} trajectories")
    prof = torch.zeros(B, 16, dtype=torch.int64, device=dev)
    lib = load()
    lib.curobo_hip_rollout_fused_set_profile_buffer(prof.data_ptr())
    ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    lib.curobo_hip_rollout_fused_set_profile_buffer(None)
    t = prof.cpu().numpy().astype(np.float64) / 100.0  # us
    names = ["P0 tables+spline", "P1 FK+spheres", "P2 costs+VJP", "P3 spline VJP"]
    for i, n in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"  {n:18s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  max {d.max():7.2f}")
    for lo, hi, n in ((1, 8, "  P1: sin/cos     "), (8, 9, "  P1: barrier + quad chains"), (9, 10, "  P1: barrier + spheres"),
                      (10, 2, "  P1: barrier wait"),
                      (2, 5, "  P2: self (pt b%H)"), (5, 6, "  P2: scene"), (2, 7, "  P2: rows (slowest)"), (7, 15, "  P2: leftover point"), (2, 12, "  P2: collision pass"),
                      (12, 13, "  P2: pose pass"), (13, 14, "  P2: c-space pass"), (14, 3, "  P2: gather pass")):
        d = t[:, hi] - t[:, lo]
        d = d[np.abs(d) < 1e6]  # rows whose stamped point was a leftover point carry no inner stamps
        print(f"  {n:18s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  max {d.max():7.2f}")
    first_end = t[:, 4].min()
    print(f"  workgroups started before the first one finished: {(t[:, 0] < first_end).sum()}")
    st = np.sort(t[:, 0] - t[:, 0].min())
    print("  start-time deciles (us):", np.round(st[:: max(1, len(st) // 10)], 1))
    tot = t[:, 4] - t[:, 0]
    # what a different dispatch order could buy: greedy list scheduling of the measured workgroup
    # durations on the slots that were concurrently busy (2 per CU), in index order vs longest first
    import heapq

    def makespan(durs, slots=512):
        heap = [0.0] * slots
        for d in durs:
            heapq.heappush(heap, heapq.heappop(heap) + d)
        return max(heap)
    print(f"  list-schedule makespan on 512 slots: index order {makespan(tot):.1f} us, longest-first "
          f"{makespan(np.sort(tot)[::-1]):.1f} us, lower bound {tot.sum() / 512:.1f} us; "
          f"duration deciles {np.round(np.percentile(tot, [10, 50, 90, 99, 100]), 1)}")
    print(f"  workgroup total    mean {tot.mean():7.2f} us; kernel span {t[:, 4].max() - t[:, 0].min():.1f} us; "
          f"concurrent workgroups ~{tot.sum() / (t[:, 4].max() - t[:, 0].min()):.0f}")


if __name__ == "__main__":
    main()
