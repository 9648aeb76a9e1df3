#!/usr/bin/env python
"""bench.py -- trajopt rollouts/sec on MI355X (BASELINE.json metric, config C2).

A *step* is one L-BFGS iteration of the reference's trajectory optimiser over one batch of
synthetic seeds: generate the 4 line-search candidates per seed, evaluate cost
AND gradient of every candidate trajectory (B-spline -> FK -> self + swept scene collision ->
per-trajectory sum -> FK backward -> B-spline backward), run the Wolfe line search and the fused
L-BFGS direction update.  Workload (per GPU): Franka Panda, 256 seeds x 4 line-search
candidates = 1024 rollouts of 32 steps (padded 33) per step, 4-cuboid world.  Rollouts/s counts
cost+gradient trajectory evaluations.  With --gpus N the seed axis shards (weak scaling: every
rank owns 256 seeds of its own, no data-path collective); the only exchange is the RCCL
all-gather arg-min over seeds at the end of the timed region.

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` (dominant kernel, live
HIP-event timing) and `cpu_baseline` (the C oracle on the host cores, bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seeds", type=int, default=256, help="seeds per GPU")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--graph-iters", type=int, default=25,
                    help="L-BFGS iterations per captured graph (the reference's inner_iters, lbfgs_bspline_trajopt.yml)")
    ap.add_argument("--no-fused", action="store_true",
                    help="drop-in kernel sequence (7 launches per rollout) instead of the fused rollout kernel")
    ap.add_argument("--shards", type=int, default=4,
                    help="seed shards of the optimiser on separate HIP streams of the GPU (1 = one batch, one stream)")
    ap.add_argument("--no-ik", action="store_true", help="skip the secondary IK solves/s measurement")
    ap.add_argument("--ik-problems", type=int, default=100)
    ap.add_argument("--ik-seeds", type=int, default=64)
    return ap.parse_args()


def time_kernel(fn, iters, torch):
    """Average duration (us) of `fn` (one kernel launch on the current stream) with HIP events."""
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model, scene_arrays, cfg, knots, start, budget_s):
    """The CPU oracle (restatement of the reference kernels) on the host cores of this box."""
    from oracle import load_oracle
    from oracle.rollout_ref import rollout_cost_and_gradient

    orc = load_oracle()
    cores = usable_cores()
    orc.set_num_threads(cores)
    sample = knots[:64]
    kw = dict(interpolation_steps=cfg.interpolation_steps, degree=cfg.bspline_degree, traj_dt=cfg.traj_dt,
              self_collision_weight=cfg.self_collision_weight, scene_collision_weight=cfg.scene_collision_weight,
              activation_distance=cfg.activation_distance, use_sweep=cfg.use_sweep,
              use_speed_metric=cfg.use_speed_metric)
    md = model.as_dict()
    rollout_cost_and_gradient(orc, md, scene_arrays, sample, start, **kw)  # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        rollout_cost_and_gradient(orc, md, scene_arrays, sample, start, **kw)
        n += sample.shape[0]
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {
        "value": n / el, "unit": "rollouts/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"{sample.shape[0]} trajectories x {cfg.padded_horizon} points per pass, cost+grad, "
                  f"{n // sample.shape[0]} passes in {el:.1f} s, OpenMP over points ({orc.num_threads()} threads)",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    local_rank %= torch.cuda.device_count()  # (only differs in the single-GPU gloo smoke test below)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # CUROBO_BENCH_BACKEND=gloo lets two ranks share one GPU to smoke-test the multi-rank logic
        backend = os.environ.get("CUROBO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from curobo_amd import _lib
    from curobo_amd.distributed import global_argmin
    from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    _lib.load()  # fail loudly if the HIP library is missing
    model = load_packaged_robot("franka")
    kin = KinematicsParams.from_model(model, device)
    scene_arrays = cuboid_scene_arrays(c2_world())
    scene = SceneData.from_arrays(scene_arrays, device)
    cfg = CollisionRolloutCfg(use_fused=not args.no_fused)
    ocfg = LBFGSOptCfg(num_problems=args.seeds, inner_iters=args.graph_iters)
    nls = len(ocfg.line_search_scale)
    rollout = CollisionRollout(kin, scene, args.seeds * nls, cfg)
    start = start_configuration(model)
    rollout.update_start_state(torch.as_tensor(start, device=device))
    bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
    start_t = torch.as_tensor(start, device=device)
    if args.shards > 1:
        # the seeds are independent problems: shard them over HIP streams so that the optimiser-side
        # kernel of one shard overlaps the rollout workgroups of the others (optim/pipelined.py)
        from curobo_amd.optim import PipelinedLBFGS

        def shard_rollout(batch):
            ro = CollisionRollout(kin, scene, batch, cfg)
            ro.update_start_state(start_t)
            return ro.cost_and_gradient
        opt = PipelinedLBFGS(ocfg, shard_rollout, cfg.n_knots, kin.num_dof, bounds, device, n_shards=args.shards,
                             use_cuda_graph=not args.no_graph)
    else:
        opt = LBFGSOpt(ocfg, rollout.cost_and_gradient, cfg.n_knots, kin.num_dof, bounds, device,
                       use_cuda_graph=not args.no_graph)
    opt1 = None
    knots = seed_knots(model, args.seeds, cfg.n_knots, seed=2, seed_offset=rank * args.seeds)
    seed_t = torch.as_tensor(knots, device=device)
    opt.reinitialize(seed_t)

    G = args.graph_iters
    rem_graphs = {}

    def run_steps(k):
        """exactly k optimiser iterations"""
        nonlocal opt1
        one = opt.step if args.shards > 1 else opt._opt_step
        if args.no_graph:
            for _ in range(k):
                one()
            return
        for _ in range(k // G):
            opt.run_inner()
        if k % G:  # remainder: its own (cached) graph, so any step count runs at replay speed
            if k % G not in rem_graphs:
                rem_graphs[k % G] = opt.make_graph(k % G)
            rem_graphs[k % G].replay()

    run_steps(max(args.warmup, 1))
    if not args.no_graph:  # every graph the timed region replays is captured before it
        if args.steps >= G and opt._graph is None:
            opt.capture()
        if args.steps % G and args.steps % G not in rem_graphs:
            rem_graphs[args.steps % G] = opt.make_graph(args.steps % G)
    # warm the exchange too (first use of the torch index/min kernels loads their code objects)
    global_argmin(opt.best_cost.view(1, -1), opt.best_action.view(1, args.seeds, -1), rank * args.seeds)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    # the one real exchange of the path: arg-min over the seeds of all ranks (1 problem)
    best_c, best_i, best_x = global_argmin(opt.best_cost.view(1, -1), opt.best_action.view(1, args.seeds, -1),
                                           rank * args.seeds)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    rollouts_per_step = args.seeds * nls * world
    value = rollouts_per_step * args.steps / elapsed

    # ---------------- per-kernel live timing (HIP events on the launch stream) -> roofline
    out = None
    if rank == 0:
        B, H = rollout.batch_size, cfg.padded_horizon
        N = B * H
        D, T, L, S = kin.num_dof, kin.num_pose_links, kin.num_links, kin.num_spheres
        act = opt.x_set.view(B, cfg.n_knots, D)
        rollout.evaluate_action(act)
        rollout.backward()
        it = 200
        kernels = {
            "bspline_forward": (lambda: rollout.compute_state_from_action(act), B * (cfg.n_knots * D * 4 + 4 * H * D * 4)),
            "fk_forward_spheres": (lambda: rollout.compute_kinematics(rollout.position), N * (4 * D + 28 * T + 16 * S + 48 * L)),
            "self_collision": (lambda: rollout_self(rollout), N * (16 * S + 4)),
            "scene_collision_swept": (lambda: rollout_scene(rollout), N * (36 * S)),
            "fk_backward": (lambda: rollout_bwd_fk(rollout), N * (48 * L + 16 * S + 28 * T + 4 * D)),
        }
        fused = cfg.use_fused and rollout.fused_available()
        if fused:  # one launch does the work of the five kernels above + cost sum + B-spline VJP
            kernels["rollout_trajectory_fused"] = (lambda: rollout.cost_and_gradient_fused(act),
                                                   N * rollout.algorithmic_bytes_per_point())
        timings = {}
        for name, (fn, nbytes) in kernels.items():
            us = time_kernel(fn, it, torch)
            timings[name] = {"us": round(us, 2), "algorithmic_bytes": int(nbytes),
                             "GBps": round(nbytes / us * 1e-3, 1)}
        step_us = time_kernel(opt.step if args.shards > 1 else opt._opt_step, 50, torch)
        dom = "rollout_trajectory_fused" if fused else max(timings, key=lambda k: timings[k]["us"])
        ach = timings[dom]["GBps"]
        traffic, traffic_src = measured_traffic(dom, B)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": timings[dom]["us"], "algorithmic_bytes_per_launch": timings[dom]["algorithmic_bytes"],
                    "launch": {"trajectories": B, "concurrent_launches": 1.0,
                               "timing": "HIP events around back-to-back launches on the launch stream"}}
        if fused and not args.no_graph:
            # the launches of the timed region as they run inside the replayed hipGraph (one per seed
            # shard and iteration, overlapping across the shard streams): device wall-clock stamps
            try:
                in_graph = graph_launch_times(opt, args, nls, torch)
            except Exception as e:  # noqa: BLE001  (keep the HIP-event figures of the exclusive launch)
                in_graph = None
                roofline["in_graph_timing_error"] = f"{type(e).__name__}: {e}"
        if fused and not args.no_graph and in_graph is not None:
            shard_rows = B // args.shards
            nbytes = shard_rows * H * rollout.algorithmic_bytes_per_point()
            ach = nbytes / in_graph["avg_launch_us"] * 1e-3
            traffic, traffic_src = measured_traffic(dom, shard_rows)
            exclusive = {k: roofline[k] for k in ("achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "launch")}
            roofline.update({
                "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": in_graph["avg_launch_us"], "algorithmic_bytes_per_launch": int(nbytes),
                "launch": {"trajectories": shard_rows, "concurrent_launches": in_graph["concurrency"],
                           "timing": "device wall-clock stamps (100 MHz) of the launches inside the replayed hipGraph"},
                "aggregate_achieved": round(ach * in_graph["concurrency"], 1),
                "exclusive_launch": exclusive})
        total_bytes = N * rollout.algorithmic_bytes_per_point()
        out = {
            "metric": "trajopt rollouts/sec (batch x horizon cost+grad)",
            "value": round(value, 1), "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C2: Franka Panda trajopt, 256 seeds x 4 line-search candidates x 32-step horizon "
                            "(padded 33), 4-cuboid world, swept scene collision + speed metric + self collision, "
                            "one L-BFGS iteration (line search + two-loop) per step",
                "robot": "franka", "seeds_per_gpu": args.seeds, "line_search_candidates": nls,
                "horizon": cfg.horizon, "n_knots": cfg.n_knots, "rollouts_per_step_per_gpu": args.seeds * nls,
                "points_per_step_per_gpu": N, "hip_graph": not args.no_graph, "fused_rollout_kernel": bool(fused), "streams_per_gpu": args.shards,
                "parallelism": f"seed-shard x{world} GPUs x{args.shards} streams",
            },
            "roofline": roofline,
            "kernels_us": {k: v["us"] for k, v in timings.items()},
            "kernels_GBps": {k: v["GBps"] for k, v in timings.items()},
            "eager_step_us": round(step_us, 1),
            "stack_algorithmic_GBps": round(total_bytes / (elapsed / args.steps) * 1e-9, 1),
            "best_cost": float(best_c[0].item()), "best_seed": int(best_i[0].item()),
        }
        # secondary measurements never take the headline line down with them
        def guarded(key, fn):
            try:
                out[key] = fn()
            except Exception as e:  # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_ik:
            guarded("ik", lambda: ik_benchmark(args, model, kin, device, torch))
            guarded("full_trajopt_rollout", lambda: full_trajopt_benchmark(args, model, kin, scene, device, torch))
            guarded("trajopt_solve", lambda: trajopt_solve_benchmark(model, kin, scene, device, torch))
        if world == 1 and not args.no_cpu_baseline:
            guarded("cpu_baseline", lambda: cpu_baseline(model, scene_arrays, cfg, knots, start, args.cpu_seconds))
            if "value" in out["cpu_baseline"]:
                out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def graph_launch_times(opt, args, nls, torch):
    """Average duration of the fused rollout launches INSIDE the replayed hipGraph: the graph is
    re-captured with the library's profile sequence on (every launch stamps its workgroups' start /
    end wall clock into its own block), replayed, and the stamps of the last replay are read back.
    concurrency = sum of launch durations / time during which any of them was running."""
    from curobo_amd._lib import load

    lib = load()
    G, shards = args.graph_iters, max(args.shards, 1)
    rows = args.seeds // shards * nls
    warm = shards  # capture() runs one eager warm-up iteration per shard first: those blocks are skipped
    buf = torch.zeros((warm + shards * G, rows, 16), dtype=torch.int64, device="cuda")
    lib.curobo_hip_rollout_fused_set_profile_sequence(buf.data_ptr(), buf.shape[0], rows)
    opt._graph = None
    opt.capture()
    lib.curobo_hip_rollout_fused_set_profile_sequence(None, 0, 0)
    for _ in range(3):
        opt.run_inner()
    torch.cuda.synchronize()
    t = buf[warm:].cpu().numpy().astype(np.float64) / 100.0  # us
    start, end = t[:, :, 0].min(axis=1), t[:, :, 4].max(axis=1)
    dur = end - start
    # union of the launch intervals
    order = np.argsort(start)
    busy, cur_s, cur_e = 0.0, start[order[0]], end[order[0]]
    for i in order[1:]:
        if start[i] > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = start[i], end[i]
        else:
            cur_e = max(cur_e, end[i])
    busy += cur_e - cur_s
    opt._graph = None  # the instrumented graph is not reused
    return {"avg_launch_us": round(float(dur.mean()), 2), "concurrency": round(float(dur.sum() / busy), 2),
            "launches": int(len(dur)), "rollout_busy_us_per_step": round(float(busy / G), 2)}


def measured_traffic(kernel: str, trajectories: int = 0):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, see profiles/*_pmc_fused.json);
    counters cannot be collected from inside the timed process, so the figure is read back from
    the newest matching profile (None if there is none for this kernel)."""
    import glob

    best = (None, None)
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_*.json"))):
        try:
            with open(path) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        if kernel not in str(rec.get("kernel", "")):
            continue
        by_shape = rec.get("hbm_bytes_per_launch_by_trajectories", {})
        if str(trajectories) in by_shape:
            best = (int(by_shape[str(trajectories)]), "profiles/" + os.path.basename(path))
        elif not by_shape and rec.get("hbm_bytes_per_launch") and trajectories in (0, 1024):
            best = (int(rec["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(path))
    return best


def full_trajopt_benchmark(args, model, kin, scene, device, torch):
    """Secondary: the FULL reference trajopt cost set (tool pose + c-space STATE + self + swept
    scene collision, lbfgs_bspline_trajopt.yml) on the C2 shapes: one fused launch vs the
    ten-launch kernel sequence, cost + gradient of 1024 trajectories."""
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.workloads import seed_knots, start_configuration

    B = args.seeds * 4
    res = {}
    knots = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=device)
    for name, fused in (("fused_us", True), ("kernel_sequence_us", False)):
        ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(use_fused=fused))
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        x = knots.reshape(B, -1)
        ro.cost_and_gradient(x)
        torch.cuda.synchronize()
        g, reps = torch.cuda.CUDAGraph(), 10  # hipGraph replay: device time, not Python launch overhead
        with torch.cuda.graph(g):
            for _ in range(reps):
                ro.cost_and_gradient(x)
        res[name] = round(time_kernel(g.replay, 20, torch) / reps, 1)
    res["rollouts_per_s_fused"] = round(B / res["fused_us"] * 1e6, 1)
    res["workload"] = "C2 shapes, full trajopt cost set (pose + c-space state + self + swept scene), cost+grad"
    return res


def trajopt_solve_benchmark(model, kin, scene, device, torch):
    """Secondary: pose-to-pose trajectory optimisation end to end (TrajOptSolver: collision-free IK with
    ranked goal configurations -> one B-spline seed per IK solution -> 100 L-BFGS iterations on the full
    trajopt cost set -> metrics and winner), C2 world, goals = FK of collision-free configurations."""
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import feasible_goals, start_configuration

    start = torch.as_tensor(start_configuration(model))
    res = {}
    for P, S in ((1, 8), (64, 4)):
        slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S))
        gp, gq = feasible_goals(kin, scene, 64)
        gp, gq = gp[:P].contiguous(), gq[:P].contiguous()
        r = slv.solve_pose(start, gp, gq)  # warm-up + graph capture
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            r = slv.solve_pose(start, gp, gq)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[f"{P}_problems_x_{S}_seeds"] = {"ms_per_batch": round(dt * 1e3, 2), "success_rate": round(float(r.success.float().mean()), 3),
                                            "lbfgs_iterations": slv.cfg.optimizer.num_iters}
    res["workload"] = ("Franka, C2 world, 32-step horizon, pose goal; IK (64 seeds, 100 iterations) + trajopt (100 iterations, "
                       "pose + c-space state + self + swept scene collision) + metrics; context only: the reference "
                       "publishes 31 ms mean solve time for its full motion planner on an RTX 6000 Ada")
    return res


def ik_benchmark(args, model, kin, device, torch):
    """Secondary metric of BASELINE.json: collision-free IK solves/s (config C1: Franka, 64 seeds
    per problem, 4-cuboid world, 100 problems per batch as in the reference's ik_benchmark.py;
    100 L-BFGS iterations of 4 line-search candidates each, hipGraph replay)."""
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import IKSolver, IKSolverCfg
    from curobo_amd.workloads import c1_world, feasible_goals

    scene = SceneData.from_arrays(cuboid_scene_arrays(c1_world()), device)
    P, S = args.ik_problems, args.ik_seeds
    shards = 4 if P % 4 == 0 else 1  # problem shards on HIP streams (optim/pipelined.py)
    solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, stream_shards=shards))
    gp, gq = feasible_goals(kin, scene, P)
    def timed(exit_early, reps=5):
        res = solver.solve_pose(gp, gq, exit_early=exit_early)  # warm-up (+ graph capture the first time)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ran = 0
        for _ in range(reps):
            res = solver.solve_pose(gp, gq, exit_early=exit_early)
            ran += int(solver.optimizer_ran)
        torch.cuda.synchronize()
        return res, (time.perf_counter() - t0) / reps, ran

    res_full, dt_full, _ = timed(False)  # every solve runs the 100 L-BFGS iterations
    res, dt, ran = timed(True)  # the reference's benchmark setting (ik_benchmark.py: config.exit_early = True)
    ocfg = solver.cfg.optimizer
    stat = lambda r: {  # noqa: E731
        "success_rate": round(float(r.success.float().mean().item()), 4),
        "median_position_error_m": float(r.position_error[r.success].median().item()) if bool(r.success.any()) else None}
    return {
        "value": round(P / dt, 1), "unit": "IK solves/s", "ms_per_batch": round(dt * 1e3, 3), "problems": P,
        "seeds_per_problem": S, **stat(res), "exit_early": True, "solves_that_ran_lbfgs": f"{ran}/5",
        "full_optimizer": {"value": round(P / dt_full, 1), "ms_per_batch": round(dt_full * 1e3, 3),
                           "lbfgs_iterations": ocfg.num_iters,
                           "rollout_rows_per_s": round(P * S * len(ocfg.line_search_scale) * (ocfg.num_iters + 1) / dt_full, 1),
                           **stat(res_full)},
        "lm_seed_solver": bool(solver.cfg.use_lm_seed), "stream_shards": shards,
        "reference_published": {"ms_per_batch": 2.726, "success_rate": 1.0, "hardware": "an NVIDIA GPU the page does not name",
                                "source": "reference docs/reference/benchmarks.rst:62-72 (franka.yml, batch 100, collision-free IK)"},
        "workload": "C1: Franka 7-DoF, 64 seeds per problem (best 64 of 128 Levenberg-Marquardt seed-IK runs, as the "
                    "reference's use_lm_seed), 4-cuboid world, pose + joint-limit + self + scene collision checks, "
                    "goals = FK of rejection-sampled collision-free configurations, exit_early as in the reference's "
                    "ik_benchmark.py (L-BFGS is skipped when the seed-IK solutions pass every check for all problems); "
                    "full_optimizer = the same solve with the 100 L-BFGS iterations (4 line-search candidates each) forced",
    }


def rollout_self(r):
    from curobo_amd.backends import geometry as g

    k, sc = r.kin, r.kin.self_collision
    g.self_collision_distance(r.self_dist, r.self_grad, r._pair_distance, r.self_sparse, r.robot_spheres,
                              sc.sphere_padding, r._w_self, sc.collision_pairs, r._bbmv, r._bbmi, 1, 256,
                              r.batch_size, r.cfg.padded_horizon, k.num_spheres, sc.collision_pairs.shape[0],
                              False, True)


def rollout_scene(r):
    from curobo_amd.backends import collision as c

    c.sphere_obstacle_collision(r.scene_dist, r.scene_grad, r.robot_spheres, r.scene.struct, r._w_scene, r._eta,
                                r.env_query_idx, r.batch_size, r.cfg.padded_horizon, r.kin.num_spheres, False,
                                3 if r.cfg.use_sweep else 0, r.cfg.use_sweep and r.cfg.use_speed_metric, r._speed_dt)


def rollout_bwd_fk(r):
    from curobo_amd.backends import kinematics as kb

    k = r.kin
    kb.launch_kinematics_backward(
        r.grad_q, r.grad_zero_pos, r.grad_zero_quat, r.self_grad, r.com, r.com, r.grad_zero_pos, r.cumul_mat,
        k.link_spheres, k.link_masses_com, k.link_map, k.joint_map, k.joint_map_type, k.tool_frame_map,
        k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets, k.joint_links_data, k.joint_links_offsets,
        k.joint_affects_endeffector, k.joint_offset_map, r.env_query_idx, k.num_envs,
        r.batch_size * r.cfg.padded_horizon, r.cfg.padded_horizon, r.action_dim, k.num_spheres, False, False,
        grad_spheres_b=r.scene_grad)


if __name__ == "__main__":
    main()
