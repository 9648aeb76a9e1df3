"""One pose-to-pose trajectory optimisation (1 problem x 8 seeds, C2 world) as a GPU timeline: run under
`rocprofv3 --kernel-trace --output-format csv`, then summarise the kernels of the LAST solve.
    python tools/r05/trajopt_timeline.py run | show <csv>"""
import csv
import collections
import os
import sys

if sys.argv[1] == "run":
    import time

    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import c2_world, feasible_goals, start_configuration

    dev = torch.device("cuda:0")
    model = load_packaged_robot("franka")
    kin = KinematicsParams.from_model(model, dev)
    scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
    P, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 8)
    slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
    gp, gq = feasible_goals(kin, scene, 64)
    gp, gq = gp[:P].contiguous(), gq[:P].contiguous()
    start = torch.as_tensor(start_configuration(model))
    for _ in range(3):
        slv.solve_pose(start, gp, gq)
        torch.cuda.synchronize()
    time.sleep(0.2)
    t0 = time.perf_counter()
    slv.solve_pose(start, gp, gq)
    torch.cuda.synchronize()
    print(f"solve {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    cut = max(i for i in range(len(t) - 1) if t[i + 1][0] - t[i][1] > 150e6) + 1  # the LAST pause of 0.2 s: the solve that was timed
    last = t[cut:]
    t0, t1 = last[0][0], max(e for _, e, _ in last)
    busy = sum(e - s for s, e, _ in last)
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n in last:
        k = n.split("(")[0].replace("void ", "")[:70]
        by[k][0] += 1
        by[k][1] += e - s
    print(f"span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us ({100 * busy / (t1 - t0):.0f} %), {len(last)} launches")
    for k, (n, ns) in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"  {ns / 1e3:9.1f} us  x{n:5d}  avg {ns / n / 1e3:6.1f}  {k}")
    gaps = sorted(((last[i + 1][0] - max(x[1] for x in last[:i + 1][-3:]), i) for i in range(len(last) - 1)), reverse=True)[:8]
    print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), last[i + 1][2].split("(")[0][-40:]) for g, i in gaps])
