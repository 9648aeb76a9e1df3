"""One collision-free IK batch (C1, exit_early) as a GPU timeline: run under `rocprofv3 --kernel-trace --output-format csv`,
then pass the *_kernel_trace.csv to the second form to list the kernels of the LAST solve with gaps.
    python tools/r05/ik_timeline.py run            (target of the profiler: 5 solves)
    python tools/r05/ik_timeline.py show <csv>"""
import csv
import os
import sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import IKSolver, IKSolverCfg
    from curobo_amd.workloads import c1_world, feasible_goals

    dev = torch.device("cuda:0")
    model = load_packaged_robot("franka")
    kin = KinematicsParams.from_model(model, dev)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), dev)
    solver = IKSolver(kin, scene, 100, IKSolverCfg(num_seeds=64, stream_shards=4))
    gp, gq = feasible_goals(kin, scene, 100)
    for _ in range(5):
        solver.solve_pose(gp, gq, exit_early=True)
        torch.cuda.synchronize()
    import time
    time.sleep(0.05)
    solver.solve_pose(gp, gq, exit_early=True)  # the one the timeline shows
    torch.cuda.synchronize()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    # the last burst: kernels after the largest gap
    gaps = [(t[i + 1][0] - t[i][1], i) for i in range(len(t) - 1)]
    cut = max(gaps)[1] + 1
    last = t[cut:]
    t0 = last[0][0]
    busy = 0
    prev_end = t0
    for s, e, n in last:
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n[:90]}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {len(last)} kernels")
