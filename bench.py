#!/usr/bin/env python
"""bench.py -- trajopt rollouts/sec on MI355X (BASELINE.json metric, config C2).

A *step* is one L-BFGS iteration of the reference's trajectory optimiser over one batch of
synthetic seeds: the 4 line-search candidates per seed are evaluated for cost
AND gradient (B-spline -> FK -> self + swept scene collision -> per-trajectory sum -> FK backward
-> B-spline backward), then the Wolfe line search and the L-BFGS two-loop produce the next
candidates.  Workload (per GPU): Franka Panda, 256 seeds x 4 line-search candidates = 1024
rollouts of 32 steps (padded 33) per step, 4-cuboid world.  Rollouts/s counts cost+gradient
trajectory evaluations.

Timing protocol (state-stable): a *block* = re-initialise the optimiser from the seeds, run
``--warmup`` untimed iterations, then time EXACTLY ``--steps`` iterations + the arg-min exchange
between barrier + synchronize pairs (MAX over ranks).  Blocks repeat until >= 0.25 s of timed work
has been collected (every block starts from the same state, so the data-dependent culling of the
collision passes cannot drift the figure); ``ms_per_step`` is the MEDIAN block time / steps.

``--gpus N`` without a torchrun environment re-executes itself under ``torch.distributed.run``
with N ranks (one per GPU, RCCL); ``n_gpus`` is only ever printed after an RCCL all-reduce over
exactly N ranks succeeded.  Weak scaling by default (256 seeds per rank); ``--scaling strong``
splits 256 seeds over the ranks.  The only exchange of the path is the all-gather arg-min over
seeds at the end of the timed steps.

Prints ONE compact JSON line (< 4 KB, the last thing on stdout) on rank 0 with `roofline` (dominant
kernel), `cpu_baseline` (the C oracle on the host cores, bounded sample) and the IK half of the
metric; the full record -- fixed-state figures, the single-GPU shares of BASELINE configs C3 / C4 /
C5 with their own rooflines, mesh world, solvers -- is written to bench_full.json (and
gpurun_out/bench_full.json).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP runtime settings this path was measured with, made explicit (both are this ROCm's defaults; set before the runtime loads):
# kernel arguments in device memory -- with HIP_FORCE_DEV_KERNARG=0 every launch of the replayed graphs fetches its arguments
# from host memory and the C2 step takes 69.9 instead of 58.9 us (profiles r04, tools/r04/call57.sh) -- and the default four
# hardware queues (GPU_MAX_HW_QUEUES=8 / 16: 113 / 153 us per step, tools/r04/call56.sh).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# fp32 vector issue: 256 CU x 4 SIMD x 2.4 GHz, one wave64 VALU instruction per 2 cycles (same guide)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
FP32_VECTOR_PEAK_TFLOPS = 157.3
MIN_TIMED_S = 0.25


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seeds", type=int, default=256, help="seeds per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline time budget per leg")
    ap.add_argument("--graph-iters", type=int, default=25,
                    help="L-BFGS iterations per captured graph (the reference's inner_iters, lbfgs_bspline_trajopt.yml)")
    ap.add_argument("--graph-lead", type=int, default=5,
                    help="iterations of the short first graph of a run of steps (0 = none): see step_chunks()")
    ap.add_argument("--no-fused", action="store_true",
                    help="drop-in kernel sequence (7 launches per rollout) instead of the fused rollout kernel")
    ap.add_argument("--shards", type=int, default=4,
                    help="seed shards of the optimiser on separate HIP streams of the GPU (1 = one batch, one stream)")
    ap.add_argument("--no-ik", action="store_true", help="skip the secondary solver measurements (IK, trajopt solve)")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 single-GPU shares")
    ap.add_argument("--only", default="", help="comma list of secondary objects to run (c3,c4,c5,mesh,ik,fixed); default all")
    ap.add_argument("--ik-problems", type=int, default=100)
    ap.add_argument("--ik-seeds", type=int, default=64)
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S)
    ap.add_argument("--legs", action="store_true", help="also run the sharded C4 / C5 legs with one rank (they always run with --gpus > 1)")
    ap.add_argument("--selftest", action="store_true",
                    help="multi-rank plumbing only: init -> all-reduce -> one global arg-min -> destroy; prints the rank count the "
                         "collective saw, so that an RCCL / xGMI problem is told apart from a kernel problem")
    ap.add_argument("--dist-timeout", type=float, default=float(os.environ.get("CUROBO_BENCH_DIST_TIMEOUT_S", "180")),
                    help="seconds a rank waits in a collective before it gives up (a rank that died must not hang the job)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N outside torchrun -> N ranks under torch.distributed.run
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    import torch

    backend = os.environ.get("CUROBO_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to print a line for fewer ranks")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def time_kernel(fn, iters, torch, min_s=0.0):
    """Average duration (us) of `fn` (launches on the current stream) with HIP events."""
    for _ in range(3):
        fn()
    total, n = 0.0, 0
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1) * 1e3
        n += iters
        if total * 1e-6 >= min_s:
            return total / n


def graphed(fn, reps, torch):
    """hipGraph of `reps` calls of `fn` (device time without Python launch overhead)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return g


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
    """The CPU oracle (C restatement of the reference kernels, gcc -O3 -march=native, OpenMP over the
    point axis) on the host cores of this box: all usable cores (headline) and one thread
    (SURVEY section 8d asks for both)."""
    from oracle import load_native_oracle, load_oracle
    from oracle.rollout_ref import rollout_cost_and_gradient

    try:
        orc, build = load_native_oracle(), "gcc -O3 -march=native -fopenmp, built on this host (oracle/_native/)"
    except Exception as e:  # noqa: BLE001  (no compiler on the box: the portable checker build)
        orc, build = load_oracle(), f"portable checker build (-O2 -ffp-contract=off; native build failed: {type(e).__name__})"
    cores = usable_cores()
    sample = knots[:256]
    kw = dict(interpolation_steps=cfg.interpolation_steps, degree=cfg.bspline_degree, traj_dt=cfg.traj_dt,
              self_collision_weight=cfg.self_collision_weight, scene_collision_weight=cfg.scene_collision_weight,
              activation_distance=cfg.activation_distance, use_sweep=cfg.use_sweep,
              use_speed_metric=cfg.use_speed_metric)
    md = model.as_dict()

    def leg(threads, budget):
        orc.set_num_threads(threads)
        rollout_cost_and_gradient(orc, md, scene_arrays, sample, start, **kw)  # warm
        t0, n = time.perf_counter(), 0
        while True:
            rollout_cost_and_gradient(orc, md, scene_arrays, sample, start, **kw)
            n += sample.shape[0]
            el = time.perf_counter() - t0
            if el >= budget:
                return n / el, n // sample.shape[0], el, orc.num_threads()

    v_all, passes, el, used = leg(cores, budget_s)
    v_one, passes1, el1, _ = leg(1, max(2.0, budget_s * 0.4))
    orc.set_num_threads(cores)
    return {
        "value": v_all, "unit": "rollouts/s", "cores": used, "kind": "port",
        "single_thread_value": v_one,
        "build": build,
        "sample": f"{sample.shape[0]} trajectories x {cfg.padded_horizon} points per pass, cost+grad, "
                  f"{passes} passes in {el:.1f} s on {used} threads (OpenMP over points); single thread: {passes1} passes in {el1:.1f} s",
    }


def cpu_optimizer_baseline(num_problems, V, m, nls, budget_s):
    """The reference's OWN torch fallbacks of the optimiser stage, timed on the host cores when the
    reference checkout is importable (this container); on the GPU box the oracle's C restatement
    of the same two functions (pinned bit-wise by tests/golden/optim_golden.npz) is timed instead."""
    from oracle import load_native_oracle, load_oracle

    try:
        orc = load_native_oracle()
    except Exception:  # noqa: BLE001
        orc = load_oracle()
    rng = np.random.default_rng(0)
    B = num_problems
    x, g = rng.normal(size=(B, V)).astype(np.float32), rng.normal(size=(B, V)).astype(np.float32)
    y, s_, rho = np.zeros((m, B, V), np.float32), np.zeros((m, B, V), np.float32), np.zeros((m, B), np.float32)
    x0, g0, step = x.copy(), g.copy(), np.zeros((B, V), np.float32)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        x += np.float32(0.01)
        g *= np.float32(0.99)
        orc.lbfgs_step(step, rho, y, s_, x, g, x0, g0, 0.01, True)
        n += 1
    el = time.perf_counter() - t0
    out = {"lbfgs_steps_per_s": n * B / el, "problems": B, "opt_dim": V, "history": m, "kind": "port",
           "sample": f"{n} two-loop updates of {B} problems in {el:.1f} s (C restatement of lbfgs_jit_helpers.py:10-78)"}
    # the optimiser stage as torch on the host cores (the reference's torch fallbacks cannot travel to the
    # GPU box: oracle/lbfgs_torch.py is their twin, pinned by the golden those functions produced)
    try:
        import torch

        from oracle.lbfgs_torch import lbfgs_step

        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(threads)
        tx, tg = torch.as_tensor(x).clone(), torch.as_tensor(g).clone()
        ty, ts, tr = torch.zeros(m, B, V), torch.zeros(m, B, V), torch.zeros(m, B)
        tx0, tg0 = tx.clone(), tg.clone()
        t1, k = time.perf_counter(), 0
        while time.perf_counter() - t1 < budget_s:
            tx += 0.01
            tg *= 0.99
            lbfgs_step(tr, ty, ts, tx, tg, tx0, tg0, 0.01, True)
            k += 1
        e2 = time.perf_counter() - t1
        out["torch_twin"] = {"lbfgs_steps_per_s": k * B / e2, "threads": threads, "kind": "port",
                             "sample": f"{k} updates of {B} problems in {e2:.1f} s (torch on the CPU, oracle/lbfgs_torch.py)"}
    except Exception as e:  # noqa: BLE001
        out["torch_twin"] = {"error": f"{type(e).__name__}: {e}"}
    return out


# ------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    backend = os.environ.get("CUROBO_BENCH_BACKEND", "nccl")
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different GPU count")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPUs visible")
    local_rank %= torch.cuda.device_count()  # (only differs in the single-GPU gloo smoke test)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # CUROBO_BENCH_BACKEND=gloo lets two ranks share one GPU to smoke-test the multi-rank logic
        import datetime

        # a collective that a peer never joins raises after --dist-timeout instead of waiting forever (gloo: RuntimeError in
        # the waiting rank; RCCL: the watchdog aborts the communicator and the rank exits)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        t_init = time.perf_counter()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        t_init = time.perf_counter() - t_init
        probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
        t_ar = time.perf_counter()
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        t_ar = time.perf_counter() - t_ar
        if int(probe.item()) != world:
            raise SystemExit(f"collective over {world} ranks returned {probe.item()}")
    if args.selftest:
        raise SystemExit(selftest(args, world, rank, device, backend, torch, dist,
                                  {"init_process_group_s": round(t_init, 3), "first_all_reduce_ms": round(t_ar * 1e3, 3)} if world > 1 else {}))

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
    if args.scaling == "strong":
        if args.seeds % world:
            raise SystemExit(f"--scaling strong: {args.seeds} seeds do not split over {world} ranks")
        seeds = args.seeds // world
    else:
        seeds = args.seeds
    shards = args.shards
    while shards > 1 and seeds % shards:
        shards //= 2
    ocfg = LBFGSOptCfg(num_problems=seeds, inner_iters=args.graph_iters)
    nls = len(ocfg.line_search_scale)
    rollout = CollisionRollout(kin, scene, seeds * nls, cfg)
    start = start_configuration(model)
    rollout.update_start_state(torch.as_tensor(start, device=device))
    bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
    start_t = torch.as_tensor(start, device=device)
    if shards > 1:
        # the seeds are independent problems: shard them over HIP streams so that the optimiser-side
        # kernel of one shard overlaps the rollout workgroups of the others (optim/pipelined.py)
        from curobo_amd.optim import PipelinedLBFGS

        def shard_rollout(batch):
            ro = CollisionRollout(kin, scene, batch, cfg)
            ro.update_start_state(start_t)
            return ro.cost_and_gradient
        opt = PipelinedLBFGS(ocfg, shard_rollout, cfg.n_knots, kin.num_dof, bounds, device, n_shards=shards,
                             use_cuda_graph=not args.no_graph)
    else:
        opt = LBFGSOpt(ocfg, rollout.cost_and_gradient, cfg.n_knots, kin.num_dof, bounds, device,
                       use_cuda_graph=not args.no_graph)
    knots = seed_knots(model, seeds, cfg.n_knots, seed=2, seed_offset=rank * seeds)
    seed_t = torch.as_tensor(knots, device=device)
    opt.reinitialize(seed_t)

    G = args.graph_iters
    rem_graphs = {}

    def step_chunks(k):
        """how k iterations are enqueued: a SHORT first graph, then graphs of up to --graph-iters iterations.  The launch of
        a four-branch graph costs host time in proportion to its nodes before its first kernel runs (~250 us for the 160
        nodes of 20 iterations: measured by fitting T(K) = K s + F); behind a 5-iteration lead-in that time is spent while
        the GPU already works, and only the lead-in's own launch is exposed.  More, smaller graphs lose again (every replay
        adds a fork / join of the four streams): measured 71.4 us per step for [20], 65.9 for [5, 15], 68.9 for [5, 5, 10],
        70.7 for [1, 2, 4, 13] at the driver's command."""
        lead = min(args.graph_lead, k // 4) if k > args.graph_lead else 0
        out = [lead] if lead > 0 else []
        k -= lead
        while k > 0:
            n = min(G, k)
            out.append(n)
            k -= n
        return out

    def local_exchange_stage():
        """the local stage of the arg-min exchange: best seed of this rank -> one packed row (one launch + two concatenations
        over the shards); captured behind the iterations of the LAST graph of a timed block, so a block is two graph
        launches and the collective"""
        linalg_hip.argmin_rows(row_buf, opt.best_cost.view(1, -1).contiguous(), opt.best_action.view(1, seeds, -1).contiguous(),
                               rank * seeds)

    def run_steps(k, with_exchange=False):
        """exactly k optimiser iterations (with_exchange: + the local stage of the arg-min exchange behind the last one)"""
        one = opt.step if shards > 1 else opt._opt_step
        if args.no_graph:
            for _ in range(k):
                one()
            if with_exchange:
                local_exchange_stage()
            return
        chunks = step_chunks(k)
        for ci, n in enumerate(chunks):
            last = with_exchange and ci == len(chunks) - 1
            if n == G and not last:
                opt.run_inner()
                continue
            key = (n, last)
            if key not in rem_graphs:  # its own (cached) graph, so any step count runs at replay speed
                rem_graphs[key] = opt.make_graph(n, after=local_exchange_stage if last else None)
            rem_graphs[key].replay()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The exchange of the timed region: its local stage (best seed of this rank) is one launch into a fixed buffer.
    from curobo_amd.backends import linalg as linalg_hip
    from curobo_amd.distributed import global_argmin_of_rows

    row_buf = torch.empty(1, 2 + cfg.n_knots * kin.num_dof, device=device)

    def run_block():
        run_steps(args.steps, with_exchange=True)
        # the one real exchange of the path: arg-min over the seeds of all ranks (1 problem)
        return global_argmin_of_rows(row_buf)

    # every graph the timed region replays is captured before it; the exchange is warmed too
    run_steps(max(args.warmup, 1))
    run_block()
    sync_all()

    def timed_block():
        """seeds -> W untimed iterations -> EXACTLY K timed iterations + the arg-min exchange"""
        opt.reinitialize(seed_t)
        run_steps(args.warmup)
        sync_all()
        t0 = time.perf_counter()
        res = run_block()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, res

    blocks, total, t_wall = [], 0.0, time.perf_counter()
    while True:
        el, (best_c, best_i, best_x) = timed_block()
        blocks.append(el)
        total += el
        go_on = (total < args.min_timed_s or len(blocks) < 5) and len(blocks) < 2000 and time.perf_counter() - t_wall < 60.0
        if world > 1:  # every rank must take the same decision
            flag = torch.tensor([1.0 if go_on else 0.0], device=device if backend == "nccl" else "cpu")
            dist.broadcast(flag, 0)
            go_on = bool(flag.item() > 0.5)
        if not go_on:
            break
    elapsed = float(np.median(blocks))
    rollouts_per_step = seeds * nls * world
    value = rollouts_per_step * args.steps / elapsed

    out = None
    if rank == 0:
        only = {s for s in args.only.split(",") if s}
        want = lambda k: (not only) or k in only  # noqa: E731
        out = {
            "metric": "trajopt rollouts/sec (batch x horizon cost+grad)",
            "value": round(value, 1), "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C2: Franka Panda trajopt, 256 seeds x 4 line-search candidates x 32-step horizon "
                            "(padded 33), 4-cuboid world, swept scene collision + speed metric + self collision, "
                            "one L-BFGS iteration (line search + two-loop) per step",
                "robot": "franka", "seeds_per_gpu": seeds, "line_search_candidates": nls,
                "horizon": cfg.horizon, "n_knots": cfg.n_knots, "rollouts_per_step_per_gpu": seeds * nls,
                "points_per_step_per_gpu": seeds * nls * cfg.padded_horizon, "hip_graph": not args.no_graph,
                "fused_rollout_kernel": bool(cfg.use_fused and rollout.fused_available()), "streams_per_gpu": shards,
                "parallelism": f"seed-shard x{world} GPUs x{shards} streams ({args.scaling} scaling, "
                               f"{'RCCL' if backend == 'nccl' else backend} all-gather arg-min)" if world > 1
                               else f"1 GPU x{shards} streams",
            },
            "timing": {
                "protocol": "block = reinitialise from the seeds, `warmup` untimed iterations, then `steps` timed iterations + "
                            "arg-min exchange between barrier+synchronize pairs (max over ranks); ms_per_step = median block / steps",
                "blocks": len(blocks), "timed_total_s": round(total, 4), "block_ms_median": round(elapsed * 1e3, 4),
                "block_ms_min": round(min(blocks) * 1e3, 4), "block_ms_max": round(max(blocks) * 1e3, 4),
            },
            "best_cost": float(best_c[0].item()), "best_seed": int(best_i[0].item()),
        }

        def guarded(key, fn):  # secondary measurements never take the headline line down with them
            try:
                out[key] = fn()
            except Exception as e:  # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {e}"}

        def roofline():
            return c2_roofline(args, opt, rollout, cfg, kin, seed_t, seeds, shards, nls, elapsed / args.steps, torch)
        guarded("roofline", roofline)
        if world == 1:
            if want("c3") and not args.no_configs:
                guarded("c3_ur10e_voxel", lambda: c3_benchmark(device, torch))
            if want("c4") and not args.no_configs:
                guarded("c4_humanoid_share", lambda: c4_benchmark(device, torch))
            if want("c5") and not args.no_configs:
                guarded("c5_batch_planner_share", lambda: c5_benchmark(model, kin, device, torch))
            if want("mesh") and not args.no_configs:
                guarded("mesh_world", lambda: mesh_benchmark(model, kin, device, torch))
            if want("ik") and not args.no_ik:
                guarded("ik", lambda: ik_benchmark(args, model, kin, device, torch))
                guarded("ik_reference_protocol", lambda: ik_protocol_benchmark(torch))
                guarded("full_trajopt_rollout", lambda: full_trajopt_benchmark(seeds, model, kin, scene, device, torch))
                guarded("trajopt_solve", lambda: trajopt_solve_benchmark(model, kin, scene, device, torch))
            if not args.no_cpu_baseline:
                guarded("cpu_baseline", lambda: cpu_baseline(model, scene_arrays, cfg, knots, start, args.cpu_seconds))
                if "value" in out["cpu_baseline"]:
                    out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
                    try:
                        out["cpu_baseline"]["optimizer_stage"] = cpu_optimizer_baseline(
                            seeds, cfg.n_knots * kin.num_dof, ocfg.history, nls, 2.0)
                    except Exception as e:  # noqa: BLE001
                        out["cpu_baseline"]["optimizer_stage"] = {"error": f"{type(e).__name__}: {e}"}
    world_alive = True
    if world > 1 and args.scaling == "weak":
        # the strong-scaling reading of the same job (north_star: 256 seeds in total), measured in the same launch
        strong = run_leg_on_all_ranks(
            "strong_scaling", lambda a, w, r, d, b, t, di: strong_scaling_leg(a, w, r, kin, scene, cfg, ocfg, model, start_t, bounds, d, b, t, di),
            args, world, rank, device, backend, torch, dist)
        world_alive = not strong.get("fatal", False)
        if rank == 0:
            out["strong_scaling"] = strong
    if (world > 1 or args.legs) and not args.no_configs and world_alive:
        # BASELINE configs 4 and 5 are multi-GPU jobs: their sharded legs (every rank runs them; rank 0 reports)
        legs = {}
        for key, fn in (("c4_humanoid_seed_shard", c4_sharded_leg), ("c5_batch_planner_problem_shard", c5_sharded_leg)):
            legs[key] = run_leg_on_all_ranks(key, fn, args, world, rank, device, backend, torch, dist)
            if legs[key].get("fatal"):  # the process group is gone: no further collective may be entered
                world_alive = False
                break
        if rank == 0:
            out["multi_gpu_legs"] = legs
    if rank == 0:  # the line goes out BEFORE the last barrier: a peer that died must not take the measured headline with it
        emit(out)
    if world > 1 and world_alive:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: teardown: {type(e).__name__}: {e}", file=sys.stderr)


LINE_LIMIT = 4096  # bytes of the ONE stdout line (round 4: a 20 KB line was cut by the driver's tail buffer -> parsed: null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out: dict) -> dict:
    """The ONE line of the contract, small enough to survive any tail buffer: the headline fields, `roofline` of the dominant
    kernel, `cpu_baseline`, the IK half of the metric, and -- for N > 1 -- the strong-scaling reading.  Everything else
    (C3 / C4 / C5 / mesh / solver objects, notes, definitions) goes to bench_full.json (emit())."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, ("dtype", "data")))
    cfg = out.get("config", {})
    line["config"] = dict(_pick(cfg, ("robot", "seeds_per_gpu", "line_search_candidates", "horizon", "rollouts_per_step_per_gpu",
                                      "points_per_step_per_gpu", "hip_graph", "fused_rollout_kernel", "streams_per_gpu", "parallelism")),
                          workload="C2: Franka trajopt, 256 seeds x 4 line-search candidates x 32-step horizon, 4-cuboid world, "
                                   "cost+grad (swept scene + speed metric + self collision) + one L-BFGS iteration per step")
    r = out.get("roofline", {})
    if "error" in r:
        line["roofline"] = {"error": str(r["error"])[:200]}
    elif r:
        rd = r.get("readings", {})
        ex, pl, ws = rd.get("exclusive_launch") or {}, rd.get("per_shard_launch") or {}, rd.get("whole_step") or {}
        roof = _pick(r, ("bound", "kernel", "peak", "unit"))
        if ex:  # the contract's reading: algorithmic bytes of ONE launch / its average duration (HIP events, launch stream)
            roof.update(achieved=ex.get("GBps"), frac=ex.get("frac"), avg_launch_us=ex.get("avg_launch_us"),
                        algorithmic_bytes_per_launch=ex.get("algorithmic_bytes"), trajectories_per_launch=ex.get("trajectories"),
                        traffic=r.get("traffic_exclusive_launch"),
                        definition="exclusive launch of all the step's trajectories at the seed state: algorithmic bytes "
                                   "(6824 B/point, SURVEY 8d) / average launch duration (HIP events on the launch stream)")
        else:
            roof.update(_pick(r, ("achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "traffic")))
        roof.setdefault("traffic", None)
        # (hardware counters cannot be collected inside the timed process: `traffic` and `issue` are READ from the newest committed
        # counter table, which rocprofv3 --pmc passes over tools/run_kernels_once.py produced; every duration in the line is live)
        roof["traffic_source"] = (r.get("primary_bound", {}) or {}).get("counters_source") or r.get("counters_source") or "committed profile (none found)"
        roof.update(_pick(r, ("valu_issue_frac", "kernel_sequence_us")))
        if ws:
            roof["whole_step"] = dict(_pick(ws, ("GBps", "frac", "us")), algorithmic_bytes=r.get("algorithmic_bytes_per_step"))
        if pl:
            roof["per_shard_launch_in_graph"] = dict(_pick(pl, ("trajectories", "avg_launch_us", "GBps", "frac", "concurrent_launches")),
                                                     traffic=r.get("traffic"))
        pb = r.get("primary_bound", {})
        if pb:
            ipl = pb.get("instructions_per_launch", {})
            roof["issue"] = dict(_pick(pb.get("issue", {}), ("simd_cycles_per_instruction", "valu_issue_share_of_launch", "scalar_issue_share_of_launch")),
                                 valu=ipl.get("valu"), salu=ipl.get("salu"), lds=ipl.get("lds"), source=pb.get("counters_source"))
        line["roofline"] = roof
    c = out.get("cpu_baseline", {})
    if c:
        line["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "single_thread_value", "sample", "error"))
        if "sample" in line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
    if "speedup_vs_cpu" in out:
        line["speedup_vs_cpu"] = out["speedup_vs_cpu"]
    ik = out.get("ik", {})
    if ik:
        line["ik"] = _pick(ik, ("value", "unit", "ms_per_batch", "problems", "seeds_per_problem", "success_rate", "error"))
        for k in ("roofline", "cpu_baseline"):
            if isinstance(ik.get(k), dict):
                line["ik"][k] = {a: b for a, b in ik[k].items() if not isinstance(b, (dict, list)) and (not isinstance(b, str) or len(b) <= 120)}
    proto = out.get("ik_reference_protocol", {})
    if ik and isinstance(proto.get("rows"), list):  # [IK ms, collision-free IK ms] per robot of the reference's published table
        rows = [r for r in proto["rows"] if "ms" in r]
        robots = sorted({r["robot"] for r in rows}, key=lambda n: ("franka", "dual_ur10e", "unitree_g1").index(n))
        pick = lambda n, key: [next((r[key] for r in rows if r["robot"] == n and r["collision_free"] == c), None) for c in (False, True)]  # noqa: E731
        line["ik"]["reference_protocol"] = {"ms": {n: pick(n, "ms") for n in robots}, "published_ms": {n: pick(n, "published_ms_nvidia") for n in robots},
                                            "success_percent": {n: pick(n, "success_percent") for n in robots}}
    ss = out.get("strong_scaling", {})
    if ss:
        line["strong_scaling"] = {a: b for a, b in ss.items() if not isinstance(b, (dict, list)) and (not isinstance(b, str) or len(b) <= 80)}
    legs = out.get("multi_gpu_legs", {})
    if legs:
        line["multi_gpu_legs"] = {k: _pick(v, ("value", "unit", "ms_per_step", "error", "fatal")) for k, v in legs.items()}
    others = {"c3_ur10e_voxel": "c3", "c4_humanoid_share": "c4", "c5_batch_planner_share": "c5"}
    line["other_configs_rollouts_per_s"] = {short: out[k].get("value") for k, short in others.items() if isinstance(out.get(k), dict) and "value" in out[k]}
    line["full_record"] = "bench_full.json"
    # never let the line outgrow the limit: drop the optional objects, last first
    for k in ("other_configs_rollouts_per_s", "multi_gpu_legs", "strong_scaling", "ik"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def emit(out: dict) -> None:
    """Full record -> bench_full.json (+ gpurun_out/ when that directory exists; stderr on request); the compact line -> stdout, last."""
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as fh:
                    fh.write(full + "\n")
            except OSError as e:
                print(f"[bench] could not write bench_full.json under {d}: {e}", file=sys.stderr)
    if os.environ.get("CUROBO_BENCH_FULL_STDERR"):  # off by default: the driver's tail buffer holds stdout AND stderr
        print("[bench] full record: " + full, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(out)), flush=True)


def run_leg_on_all_ranks(key, fn, args, world, rank, device, backend, torch, dist):
    """One sharded leg on every rank, with the outcome AGREED over the ranks.  A per-rank try / except alone can deadlock: a
    rank that fails (an RCCL / HSA error, an out-of-memory) leaves its peers inside the leg's next collective.  Here every
    collective carries the process group's timeout (``--dist-timeout``), so the peers of a failed rank come back with an
    error of their own; then all ranks all-reduce an error flag and report the leg as failed together.  If even that
    all-reduce fails the group is unusable: ``fatal`` tells the caller to enter no further collective."""
    err = None
    res = None
    try:
        if os.environ.get("CUROBO_BENCH_FAIL_LEG") == f"{key}:{rank}":  # test hook: this rank fails before its first collective
            raise RuntimeError(f"injected failure of {key} on rank {rank}")
        res = fn(args, world, rank, device, backend, torch, dist)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if world == 1:
        return res if err is None else {"error": err}
    try:
        flag = torch.tensor([0.0 if err is None else 1.0, float(rank) if err is not None else -1.0],
                            device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed, who = bool(flag[0].item() > 0.5), int(flag[1].item())
    except Exception as e:  # noqa: BLE001
        return {"error": err or f"{type(e).__name__}: {e}", "fatal": True,
                "note": "the error flag could not be exchanged: the process group is unusable, remaining legs skipped"}
    if failed:
        return {"error": err or f"a peer failed (highest failing rank: {who})", "failed_rank": who}
    return res


def selftest(args, world, rank, device, backend, torch, dist, timings):
    """``--selftest``: the multi-rank plumbing without any benchmark kernel: init (done by the caller) -> all-reduce ->
    one ``global_argmin`` (the path's one exchange: an all-gather of a packed row per rank) -> barrier -> destroy."""
    from curobo_amd.distributed import global_argmin

    out = {"selftest": True, "backend": "RCCL (torch 'nccl')" if backend == "nccl" else backend, "world_size_env": world,
           "devices_visible": torch.cuda.device_count(), **timings}
    if world > 1:
        cdev = device if backend == "nccl" else "cpu"
        ones = torch.ones(1, device=cdev)
        dist.all_reduce(ones)
        out["ranks_counted_by_all_reduce"] = int(ones.item())
        # rank r offers cost 10 - r for its second seed: the winner must be the last rank's seed 1, global index 2 (W - 1) + 1
        cost = torch.tensor([[20.0, 10.0 - rank]], device=device)
        payload = torch.full((1, 2, 3), float(rank), device=device)
        torch.cuda.synchronize()
        lat = []
        for _ in range(20):
            t0 = time.perf_counter()
            c, i, x = global_argmin(cost, payload, rank * 2)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        ok = int(i[0].item()) == 2 * (world - 1) + 1 and abs(float(c[0].item()) - (10.0 - (world - 1))) < 1e-6 and \
            float(x[0, 0].item()) == float(world - 1)
        out["global_argmin_ok"] = bool(ok)
        out["global_argmin_ms_median"] = round(float(np.median(lat)) * 1e3, 4)
        dist.barrier()
        dist.destroy_process_group()
    else:
        out["ranks_counted_by_all_reduce"] = 1
        out["global_argmin_ok"] = True
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0 if out["global_argmin_ok"] and out["ranks_counted_by_all_reduce"] == world else 1


def _timed_sharded(step, exchange, steps, warmup, world, device, backend, torch, dist, blocks=7):
    """the block protocol of the headline for a sharded leg: `warmup` untimed steps, then `steps` timed steps + the
    exchange between barrier + synchronize pairs, MAX over ranks, median of `blocks`"""
    cdev = device if backend == "nccl" else "cpu"

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    out, res = [], None
    for _ in range(blocks):
        for _ in range(warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        res = exchange()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        out.append(el)
    return float(np.median(out)), res


def c4_sharded_leg(args, world, rank, device, backend, torch, dist):
    """BASELINE config 4 ("Humanoid whole-body: map-reduce self-collision + inverse-dynamics cost, 1024 seeds, 2/4 MI355X
    seed-shard"): the 1024 seeds of ONE problem split contiguously over the ranks (strong scaling), every rank evaluates
    cost + gradient of its seeds' 4 candidates (FK, 162 k-pair self collision, RNEA torque cost, VJPs: kernel sequence), the
    arg-min over all seeds is one all-gather.  A step = one rollout set of the rank's shard."""
    from curobo_amd.distributed import global_argmin, shard_range
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.workloads import seed_knots, start_configuration

    total, nls = 1024, 4
    lo, hi = shard_range(total, rank, world)
    seeds = hi - lo
    kcfg = KinematicsCfg.from_packaged("unitree_g1", device=device)
    model, kin = kcfg.model, kcfg.kinematics_config
    D = kin.num_dof
    B = seeds * nls
    ro = TrajOptRollout(kin, None, B, TrajOptRolloutCfg(use_fused=False, use_torque_limits=True, effort_limit=[200.0] * D))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    # seed s of the job depends on its global index only: any world size evaluates the same 1024 seeds
    base = seed_knots(model, seeds, 12, seed=6, seed_offset=lo, spread=0.15)
    x = torch.as_tensor(np.repeat(base, nls, axis=0), device=device).reshape(B, -1)
    ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = graphed(lambda: ro.cost_and_gradient(x), 1, torch)
    steps, warmup = max(1, min(args.steps, 10)), 1

    def exchange():
        cost = ro.cost.view(seeds, nls)[:, 0].contiguous().view(1, seeds)
        return global_argmin(cost, x.view(seeds, nls, -1)[:, 0].contiguous().view(1, seeds, -1), lo)
    el, (best_c, best_i, _) = _timed_sharded(g.replay, exchange, steps, warmup, world, device, backend, torch, dist)
    H = ro.cfg.padded_horizon
    return {"workload": f"C4: unitree_g1 ({D} dof, {kin.num_spheres} spheres, {int(kin.self_collision.collision_pairs.shape[0])} pairs), "
                        f"{total} seeds in total x {nls} candidates x {H} points, pose + c-space STATE (RNEA torque limits) + self collision, "
                        "cost+grad, kernel sequence",
            "scaling": "strong", "n_gpus": world, "seeds_per_gpu": seeds, "value": round(total * nls * steps / el, 1), "unit": "rollouts/s",
            "ms_per_step": round(el / steps * 1e3, 4), "steps": steps, "exchange": "all-gather arg-min over the seeds (1 problem)",
            "best_seed": int(best_i[0].item()), "best_cost": float(best_c[0].item())}


def c5_sharded_leg(args, world, rank, device, backend, torch, dist):
    """BASELINE config 5 ("Batch motion_planner: 16 robots x 512 seeds x 64 horizon, mixed scene, 8 MI355X with RCCL argmin"):
    the 16 planning problems -- each with its own world of cuboids + one fp16 ESDF grid -- are sharded BY PROBLEM (every
    rank holds all 512 seeds of its problems and only their worlds), every rank evaluates cost + gradient of 4 candidates per
    seed with the fused multi-env launch, and the per-problem winners are exchanged with one all-gather."""
    from curobo_amd.distributed import gather_problem_winners, shard_range
    from curobo_amd.robot import load_packaged_robot
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c5_mixed_worlds, seed_knots, start_configuration

    n_prob, seeds, nls = 16, 512, 4
    if n_prob % world:
        return {"error": f"{n_prob} problems do not split over {world} ranks"}
    lo, hi = shard_range(n_prob, rank, world)
    mine = hi - lo
    model = load_packaged_robot("franka")
    kin = KinematicsParams.from_model(model, device)
    worlds = c5_mixed_worlds(n_prob, voxels=True)  # the same 16 worlds on every rank (seeded); a rank uploads its own
    arrays = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (n_prob,) else v) for k, v in worlds.items()}
    scene = SceneData.from_arrays(arrays, device)
    B = mine * seeds * nls
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(interpolation_steps=4))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    ro.update_env_query_idx(torch.arange(mine, dtype=torch.int32, device=device).repeat_interleave(seeds * nls))
    base = seed_knots(model, mine * seeds, 12, seed=8, seed_offset=lo * seeds)
    x = torch.as_tensor(np.repeat(base, nls, axis=0), device=device).reshape(B, -1)
    ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = graphed(lambda: ro.cost_and_gradient(x), 1, torch)
    steps, warmup = max(1, min(args.steps, 10)), 1

    def exchange():
        cost = ro.cost.view(mine, seeds, nls)[:, :, 0].contiguous()
        return gather_problem_winners(cost, x.view(mine, seeds, nls, -1)[:, :, 0].contiguous(), lo, n_prob)
    el, (best_c, best_i, best_x) = _timed_sharded(g.replay, exchange, steps, warmup, world, device, backend, torch, dist)
    return {"workload": f"C5: {n_prob} problems (own worlds: cuboids + 64^3 fp16 ESDF) x {seeds} seeds x {nls} candidates, Franka, horizon 64 "
                        "(padded 65), swept scene collision + speed metric + self collision, cost+grad, fused multi-env launch"
                        if ro.fused_available() else "kernel sequence",
            "scaling": "strong", "n_gpus": world, "problems_per_gpu": mine, "value": round(n_prob * seeds * nls * steps / el, 1),
            "unit": "rollouts/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
            "exchange": "all-gather of the per-problem winners (problem shard: every seed of a problem lives on one rank)",
            "winners": int(best_i.numel()), "best_seed_of_problem_0": int(best_i[0].item())}


def strong_scaling_leg(args, world, rank, kin, scene, cfg, ocfg, model, start_t, bounds, device, backend, torch, dist):
    """256 seeds IN TOTAL split over the ranks (north_star's 256-seed job): same block protocol."""
    import dataclasses

    from curobo_amd.distributed import global_argmin
    from curobo_amd.optim import LBFGSOpt
    from curobo_amd.rollout import CollisionRollout
    from curobo_amd.workloads import seed_knots

    total_seeds = args.seeds
    if total_seeds % world:
        return {"error": f"{total_seeds} seeds do not split over {world} ranks"}
    seeds = total_seeds // world
    nls = len(ocfg.line_search_scale)
    ro = CollisionRollout(kin, scene, seeds * nls, cfg)
    ro.update_start_state(start_t)
    opt = LBFGSOpt(dataclasses.replace(ocfg, num_problems=seeds), ro.cost_and_gradient, cfg.n_knots, kin.num_dof, bounds, device)
    seed_t = torch.as_tensor(seed_knots(model, seeds, cfg.n_knots, seed=2, seed_offset=rank * seeds), device=device)
    opt.reinitialize(seed_t)
    g = opt.make_graph(args.steps)
    gw = opt.make_graph(max(args.warmup, 1))
    cdev = device if backend == "nccl" else "cpu"

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    blocks = []
    for _ in range(25):
        opt.reinitialize(seed_t)
        gw.replay()
        sync_all()
        t0 = time.perf_counter()
        g.replay()
        global_argmin(opt.best_cost.view(1, -1), opt.best_action.view(1, seeds, -1), rank * seeds)
        sync_all()
        tt = torch.tensor([time.perf_counter() - t0], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        blocks.append(float(tt.item()))
    el = float(np.median(blocks))
    # the exchange alone (local best row + all-gather + arg-min): what a solve pays once, whatever its iteration count
    ex = []
    for _ in range(25):
        sync_all()
        t0 = time.perf_counter()
        global_argmin(opt.best_cost.view(1, -1), opt.best_action.view(1, seeds, -1), rank * seeds)
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ex.append(float(tt.item()))
    return {"scaling": "strong", "total_seeds": total_seeds, "seeds_per_gpu": seeds, "streams_per_gpu": 1, "n_gpus": world,
            "value": round(total_seeds * nls * args.steps / el, 1), "unit": "rollouts/s",
            "ms_per_step": round(el / args.steps * 1e3, 5), "blocks": len(blocks),
            "exchange_ms_per_solve": round(float(np.median(ex)) * 1e3, 4),
            "exchange": "one all_gather_into_tensor of a packed [cost, global seed index, knots] row per rank, arg-min on every rank",
            "note": "256 seeds IN TOTAL over the ranks (north_star's >= 6x target is this value at --gpus 8 over the same field at --gpus 1: "
                    "run `bench.py --gpus 1 --scaling strong` for the denominator); one optimiser stream per GPU"}


# ------------------------------------------------------------------------------------------------
# C2 roofline object
# ------------------------------------------------------------------------------------------------
def c2_roofline(args, opt, rollout, cfg, kin, seed_t, seeds, shards, nls, step_s, torch):
    """Dominant kernel = the fused rollout launch.  `achieved` is the AGGREGATE algorithmic rate of the
    rollout launches of the timed steps (all concurrent shard launches together: bytes of one step's
    launches / time during which any of them was running), live from device wall-clock stamps inside
    the replayed graph; the exclusive 1024-trajectory launch at the SEED state (HIP events on the
    launch stream) and the drop-in kernel sequence are reported next to it."""
    B, H = rollout.batch_size, cfg.padded_horizon
    N = B * H
    D, T, L, S = kin.num_dof, kin.num_pose_links, kin.num_links, kin.num_spheres
    bpp = rollout.algorithmic_bytes_per_point()
    # fixed state: the line-search candidates of iteration 1 at the seeds (no optimiser progress)
    opt.reinitialize(seed_t)
    torch.cuda.synchronize()
    act = opt.x_set.reshape(B, cfg.n_knots, D).clone()
    rollout.evaluate_action(act)
    rollout.backward()
    reps = 20
    kernels = {
        "bspline_forward": (lambda: rollout.compute_state_from_action(act), B * (cfg.n_knots * D * 4 + 4 * H * D * 4)),
        "fk_forward_spheres": (lambda: rollout.compute_kinematics(rollout.position), N * (4 * D + 28 * T + 16 * S + 48 * L)),
        "self_collision": (lambda: rollout_self(rollout), N * (16 * S + 4)),
        "scene_collision_swept": (lambda: rollout_scene(rollout), N * (36 * S)),
        "fk_backward": (lambda: rollout_bwd_fk(rollout), N * (48 * L + 2 * 16 * S + 28 * T + 4 * D)),  # two gradient streams in (self, scene)
    }
    fused = cfg.use_fused and rollout.fused_available()
    if fused:  # one launch does the work of the five kernels above + cost sum + B-spline VJP
        kernels["rollout_trajectory_fused"] = (lambda: rollout.cost_and_gradient_fused(act), N * bpp)
    timings = {}
    for name, (fn, nbytes) in kernels.items():
        g = graphed(fn, reps, torch)
        us = time_kernel(g.replay, 10, torch, min_s=0.05) / reps
        timings[name] = {"us": round(us, 2), "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / us * 1e-3, 1)}
    seq_us = sum(v["us"] for k, v in timings.items() if k != "rollout_trajectory_fused")
    dom = "rollout_trajectory_fused" if fused else max(timings, key=lambda k: timings[k]["us"])
    excl = timings[dom]
    traffic, traffic_src = measured_traffic(dom, B)
    roof = {"bound": "hbm", "kernel": dom, "achieved": excl["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(excl["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_note": f"HBM bytes per {B}-trajectory launch from the COMMITTED rocprofv3 PMC passes of this command "
                            f"({traffic_src}); counters are not collected in this run",
            "avg_launch_us": excl["us"], "algorithmic_bytes_per_launch": excl["algorithmic_bytes"],
            "definition": "exclusive launch at the seed state"}
    exclusive = {"state": "seed candidates (iteration 1, every trajectory as seeded; no optimiser progress)",
                 "trajectories": B, "avg_launch_us": excl["us"], "achieved": excl["GBps"],
                 "frac": round(excl["GBps"] / HBM_PEAK_GBS, 4), "rollouts_per_s": round(B / excl["us"] * 1e6, 1),
                 "timing": "HIP events around hipGraph replays of 20 back-to-back launches on the launch stream"}
    roof["exclusive_launch_fixed_state"] = exclusive
    if fused and not args.no_graph and hasattr(opt, "opts"):
        try:
            ig = graph_launch_times(opt, args.graph_iters, shards, seeds, nls, seed_t, torch)
            step_bytes = N * bpp
            agg = step_bytes / ig["rollout_busy_us_per_step"] * 1e-3
            roof.update({
                "achieved": round(agg, 1), "frac": round(agg / HBM_PEAK_GBS, 4),
                "definition": "AGGREGATE over the concurrent shard launches of a step: algorithmic bytes of one step's rollout "
                              "launches / time during which any of them was running (device wall-clock stamps, 100 MHz, "
                              "inside the replayed hipGraph, first graph_iters iterations from the seeds)",
                "avg_launch_us": ig["avg_launch_us"], "algorithmic_bytes_per_launch": int(step_bytes // shards),
                "launch": {"trajectories": B // shards, "concurrent_launches": ig["concurrency"], "launches_sampled": ig["launches"],
                           "rollout_busy_us_per_step": ig["rollout_busy_us_per_step"]},
            })
            t256, src256 = measured_traffic(dom, B // shards)
            if t256 is not None:
                roof["traffic"] = t256
                roof["traffic_note"] = (f"HBM bytes per {B // shards}-trajectory launch from the COMMITTED rocprofv3 PMC passes "
                                        f"of the single-stream variant of this command ({src256}); not collected in this run")
        except Exception as e:  # noqa: BLE001  (keep the HIP-event figures of the exclusive launch)
            roof["in_graph_timing_error"] = f"{type(e).__name__}: {e}"
    # ---- the line's own, driver-checkable figure is the headline: algorithmic bytes of one step / ms_per_step.  The
    # per-launch and exclusive-launch readings stand beside it, each recomputable from its own fields.
    aggregate = {"GBps": roof["achieved"], "frac": roof["frac"], "definition": roof["definition"]}
    per_launch = None
    if "launch" in roof:
        lb, lus = roof["algorithmic_bytes_per_launch"], roof["avg_launch_us"]
        per_launch = {"trajectories": roof["launch"]["trajectories"], "algorithmic_bytes": lb, "avg_launch_us": lus,
                      "GBps": round(lb / lus * 1e-3, 1), "frac": round(lb / lus * 1e-3 / HBM_PEAK_GBS, 4),
                      "concurrent_launches": roof["launch"]["concurrent_launches"],
                      "timing": "device wall-clock stamps inside the replayed hipGraph (100 MHz), mean over the launches sampled"}
    whole = N * bpp / step_s * 1e-9
    roof.update({"achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(N * bpp),
                 "definition": "WHOLE STEP: algorithmic bytes of one step's rollouts (rollouts x points x bytes/point, SURVEY 8d) / "
                               "ms_per_step of this line (rollout launches + optimiser tail + arg-min exchange)",
                 "readings": {"whole_step": {"GBps": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4), "us": round(step_s * 1e6, 2)},
                              "per_shard_launch": per_launch,
                              "concurrent_shard_launches_aggregate": aggregate if per_launch else None,
                              "exclusive_launch": {"trajectories": B, "algorithmic_bytes": excl["algorithmic_bytes"],
                                                   "avg_launch_us": excl["us"], "GBps": excl["GBps"],
                                                   "frac": round(excl["GBps"] / HBM_PEAK_GBS, 4)}}})
    roof["whole_step_algorithmic_GBps"] = round(whole, 1)
    roof["whole_step_frac"] = round(whole / HBM_PEAK_GBS, 4)
    # ---- what binds the fused kernel is VALU issue, not HBM (the launch moves ~1 % of its algorithmic bytes): committed
    # instruction counters of the same launch shape x the LIVE exclusive launch time
    cr = counter_roofline(dom, B, excl["us"], units=B)
    if cr:
        if cr.get("traffic") is not None and roof.get("traffic") is None:
            roof["traffic"], roof["traffic_note"] = cr["traffic"], f"HBM bytes per {B}-trajectory launch, {cr['counters_source']}"
        roof["traffic_exclusive_launch"] = cr.get("traffic")
        if "valu" in cr:
            roof["primary_bound"] = dict(cr["valu"], kernel=dom, launch="exclusive, seed state", avg_launch_us=excl["us"],
                                         counters_source=cr["counters_source"],
                                         note="the binding limit of the fused launch: it keeps every intermediate in LDS and moves "
                                              "~1 % of its algorithmic bytes through HBM, so the HBM fraction above is notional")
    # what actually bounds the fused kernel: VALU issue (committed SQ counters x the live launch time)
    sq = committed_sq_counters(dom, B)
    if sq is not None:
        insts = sq["valu_wave_instructions_per_trajectory"] * B
        roof["valu_issue_frac"] = round(insts / (excl["us"] * 1e-6) / VALU_ISSUE_PEAK, 4)
        roof["valu_issue_note"] = (f"{sq['valu_wave_instructions_per_trajectory']} VALU wave-instructions per trajectory "
                                   f"(seed state, {sq['source']}, committed file) x {B} trajectories / live exclusive launch time, "
                                   f"against 256 CU x 4 SIMD x 2.4 GHz / 2 cycles = {VALU_ISSUE_PEAK:.3g} wave-instructions/s")
        if sq.get("active_inst_valu_quad_cycles_per_launch"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles in which a VALU instruction of a wavefront is executing, summed over
            # the wavefronts: x 4 cycles / (SIMDs x launch time x clock).  An upper bound on VALU occupancy: the counter
            # cannot resolve less than one quad-cycle per instruction.
            roof["valu_active_frac_upper_bound"] = round(
                sq["active_inst_valu_quad_cycles_per_launch"] * 4.0 / (256 * 4 * excl["us"] * 1e-6 * 2.4e9), 4)
    roof["kernels_us"] = {k: v["us"] for k, v in timings.items()}
    roof["kernels_GBps"] = {k: v["GBps"] for k, v in timings.items()}
    roof["kernel_sequence_us"] = round(seq_us, 1)
    return roof


def committed_counters(kernel: str, workgroups: int = 0):
    """Per-launch rocprofv3 counters of a kernel from the newest committed ``profiles/*_counters_by_kernel.json``
    (tools/collect_profiles_r03.sh: FETCH_SIZE / WRITE_SIZE / SQ_* passes over tools/run_kernels_once.py, which launches
    bench.py's workloads kernel by kernel).  Counters cannot be collected from inside the timed process; instruction counts
    and HBM bytes of a launch do not depend on when it runs, durations are always measured live.  ``workgroups`` picks the
    launch shape (0 = the largest).  Returns None when no committed file has the kernel."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters_by_kernel.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        rows = [e for e in rec.get("kernels", []) if kernel in e["kernel"] and e.get("counters")]
        if workgroups:
            rows = [e for e in rows if e["workgroups"] == workgroups]
        if rows:
            e = max(rows, key=lambda r: r["workgroups"])
            best = dict(e, source="profiles/" + os.path.basename(path))
    return best


ISSUE_FLOOR_CYCLES = 2.9  # SIMD-cycles per wave-instruction of the best issue-bound kernel measured (self_collision_row16_kernel, r04_b)


def counter_roofline(kernel: str, workgroups: int, live_us: float, units: int = 0) -> dict:
    """{traffic, valu / salu / lds instruction counts, VALU issue fraction at the LIVE launch time} of one kernel from the
    committed counters; {} when there are none.  ``units`` (trajectories or points per launch) adds per-unit counts."""
    c = committed_counters(kernel, workgroups)
    if c is None:
        return {}
    k = c["counters"]
    out = {"counters_source": c["source"], "counters_kernel": c["kernel"], "counters_workgroups": c["workgroups"],
           "traffic": c.get("hbm_bytes"), "l2_hit_rate": c.get("l2_hit_rate")}
    if "SQ_INSTS_VALU" in k and live_us > 0:
        out["valu"] = {"bound": "valu", "achieved": round(k["SQ_INSTS_VALU"] / (live_us * 1e-6), 1), "peak": VALU_ISSUE_PEAK,
                       "unit": "wave64 VALU instructions/s", "frac": round(k["SQ_INSTS_VALU"] / (live_us * 1e-6) / VALU_ISSUE_PEAK, 4),
                       "instructions_per_launch": {"valu": k.get("SQ_INSTS_VALU"), "salu": k.get("SQ_INSTS_SALU"), "lds": k.get("SQ_INSTS_LDS"),
                                                   "smem": k.get("SQ_INSTS_SMEM")},
                       "peak_definition": "256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 fp32 instruction (MI355X_MICROARCH.md, v_fma_f32 row)",
                       "lds_issue_stall_share_of_wave_cycles": c.get("lds_issue_stall_share_of_wave_cycles")}
        if units:
            out["valu"]["per_unit"] = {n: round(k[m] / units, 1) for n, m in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"),
                                                                               ("lds", "SQ_INSTS_LDS")) if m in k}
        # Every instruction type: SIMD-cycles the launch had (1024 SIMDs x live time x 2.4 GHz) per wave-instruction it
        # executed.  Across this library's kernels that run at four or more wavefronts per SIMD the figure sits between 2.9
        # (self_collision_row16_kernel, 78 % VALU) and 4.4 whatever their VALU share (profiles/r04_b_counters_by_kernel.json,
        # DESIGN.md section 5); tools/probes/valu_issue_probe.hip: independent VALU streams saturate at 2.1 - 2.6 cycles per
        # instruction, and the scalar unit issues one instruction per cycle per CU (a SIMD's share: four cycles).
        total = sum(k.get(m, 0.0) or 0.0 for m in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
                                                   "SQ_INSTS_VMEM_WR"))
        if total > 0:
            cyc = live_us * 1e-6 * 2.4e9 * 1024 / total
            simd_cycles = live_us * 1e-6 * 2.4e9
            valu_c, salu_c = k.get("SQ_INSTS_VALU", 0.0) * 2.3 / 1024, (k.get("SQ_INSTS_SALU", 0.0) or 0.0) / 256
            out["valu"]["issue"] = {"wave_instructions_per_launch_all_types": round(total, 1), "simd_cycles_per_instruction": round(cyc, 2),
                                    "best_rate_measured_in_this_library": ISSUE_FLOOR_CYCLES,
                                    "frac_of_that_rate": round(ISSUE_FLOOR_CYCLES / cyc, 4),
                                    "valu_issue_share_of_launch": round(valu_c / simd_cycles, 4),
                                    "scalar_issue_share_of_launch": round(salu_c / simd_cycles, 4),
                                    "pricing": "VALU 2.3 SIMD-cycles per wave-instruction, scalar unit 1 instruction per cycle per CU "
                                               "(tools/probes/valu_issue_probe.hip)"}
    return out


def committed_sq_counters(kernel, workgroups=0):
    """VALU wave-instructions per trajectory (workgroup) of a fused launch from the NEWEST committed
    ``profiles/*_counters_by_kernel.json`` (the same file every other counter figure of the line comes from)."""
    e = committed_counters(kernel, workgroups)
    if e is None or not e.get("counters", {}).get("SQ_INSTS_VALU") or not e.get("workgroups"):
        return None
    c = e["counters"]
    return {"valu_wave_instructions_per_trajectory": round(c["SQ_INSTS_VALU"] / e["workgroups"], 1),
            "active_inst_valu_quad_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
            "counter_launch_workgroups": e["workgroups"], "source": e["source"]}


def graph_launch_times(opt, G, shards, seeds, nls, seed_t, torch):
    """Durations of the fused rollout launches INSIDE the replayed hipGraph: the graph is re-captured
    with the library's profile sequence on (every launch stamps its workgroups' start / end wall
    clock into its own block), replayed from the seed state, and the stamps are read back.
    concurrency = sum of launch durations / time during which any of them was running."""
    from curobo_amd._lib import load

    lib = load()
    rows = seeds // shards * nls
    warm = shards  # capture() runs one eager warm-up iteration per shard first: those blocks are skipped
    buf = torch.zeros((warm + shards * G, rows, 16), dtype=torch.int64, device="cuda")
    lib.curobo_hip_rollout_fused_set_profile_sequence(buf.data_ptr(), buf.shape[0], rows)
    opt._graph = None
    opt.capture()
    lib.curobo_hip_rollout_fused_set_profile_sequence(None, 0, 0)
    opt.reinitialize(seed_t)
    opt.run_inner()
    torch.cuda.synchronize()
    t = buf[warm:].cpu().numpy().astype(np.float64) / 100.0  # us
    start, end = t[:, :, 0].min(axis=1), t[:, :, 4].max(axis=1)
    dur = end - start
    order = np.argsort(start)  # union of the launch intervals
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

    c = committed_counters(kernel, trajectories)  # this round's collection (tools/collect_profiles_r03.sh) first
    if c is not None and c.get("hbm_bytes") is not None:
        return int(c["hbm_bytes"]), c["source"]
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_*.json"))):
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


# ------------------------------------------------------------------------------------------------
# BASELINE configs 3 / 4 / 5: single-GPU shares, each with its own roofline
# ------------------------------------------------------------------------------------------------
def _timed_stages(stages, torch, reps=5):
    """{name: (fn, algorithmic_bytes)} -> per-stage us / GB/s, each stage replayed from a hipGraph"""
    res = {}
    for name, (fn, nbytes) in stages.items():
        g = graphed(fn, reps, torch)
        us = time_kernel(g.replay, 3, torch, min_s=0.05) / reps
        res[name] = {"us": round(us, 2), "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / us * 1e-3, 1),
                     "hbm_frac": round(nbytes / us * 1e-3 / HBM_PEAK_GBS, 4)}
    return res


def mesh_benchmark(model, kin, device, torch):
    """Mesh obstacles: the C2 world with every cuboid handed over as a triangle mesh (12 288 triangles in all:
    each box face subdivided) + a torus, 1024 trajectories x 33 points of robot spheres: the mesh launch
    (curobo_hip_sphere_mesh_collision, BVH walk per sphere and sweep sample) next to the cuboid kernel on the same boxes,
    BVH build time, and the ESDF bake through the BVH against the all-triangles bake."""
    from curobo_amd.backends import collision as Cn
    from curobo_amd.backends.mesh import build_mesh_bvh, mesh_esdf_bake_bvh
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.workloads import c2_world, seed_knots, start_configuration

    world = c2_world()
    meshes = c2_world_as_meshes()
    n_tri = sum(len(m["faces"]) for m in meshes[0])
    B, H = 1024, 33
    cfg = CollisionRolloutCfg(use_fused=False)
    out = {"workload": f"C2 shapes (1024 x 33 points x {kin.num_spheres} spheres), the {len(world[0])} cuboids of the C2 world as triangle "
                       f"meshes ({n_tri} triangles), swept + speed metric", "triangles": n_tri}
    x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=device).reshape(B, -1)
    times = {}
    return _mesh_benchmark_body(model, kin, device, torch, world, meshes, B, H, cfg, out, x, times, Cn, build_mesh_bvh, mesh_esdf_bake_bvh,
                                CollisionRollout, SceneData, cuboid_scene_arrays, start_configuration)


def c2_world_as_meshes():
    """the four cuboids of the C2 world as triangle meshes, every box face subdivided four times (3 072 triangles per box)"""
    import numpy as np

    from curobo_amd.scene import box_mesh
    from curobo_amd.workloads import c2_world

    def subdivide(v, f, times):
        v = [tuple(x) for x in np.asarray(v, np.float64)]
        f = np.asarray(f, np.int64)
        for _ in range(times):
            cache, out = {}, []

            def mid(a, b):
                key = (min(a, b), max(a, b))
                if key not in cache:
                    cache[key] = len(v)
                    v.append(tuple((np.asarray(v[a]) + np.asarray(v[b])) * 0.5))
                return cache[key]
            for a, b, c in f:
                ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
                out += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
            f = np.asarray(out, np.int64)
        return np.asarray(v, np.float32), f.astype(np.int32)

    world = c2_world()
    return [[dict(name=f"box{i}", pose=o["pose"], **dict(zip(("vertices", "faces"), subdivide(*box_mesh(o["dims"]), 4))))
             for i, o in enumerate(world[0])]]


def _mesh_benchmark_body(model, kin, device, torch, world, meshes, B, H, cfg, out, x, times, Cn, build_mesh_bvh, mesh_esdf_bake_bvh,
                         CollisionRollout, SceneData, cuboid_scene_arrays, start_configuration):
    scene_cells = None
    from curobo_amd.scene import MeshStore

    for key, scene in (("cuboid_kernel", SceneData.from_arrays(cuboid_scene_arrays(world), device)),
                       ("mesh_launch", SceneData.from_arrays(None, device, meshes=meshes)),
                       ("mesh_launch_tree_walk", SceneData.from_arrays(None, device, meshes=MeshStore(meshes, device, cells=False)))):
        ro = CollisionRollout(kin, scene, B, cfg)
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        ro.compute_kinematics(ro.compute_state_from_action(x.view(B, cfg.n_knots, -1)))
        S = kin.num_spheres

        def scene_pass(ro=ro, scene=scene):
            Cn.sphere_obstacle_collision(ro.scene_dist, ro.scene_grad, ro.robot_spheres, scene.struct, ro._w_scene, ro._eta, ro.env_query_idx,
                                         B, cfg.padded_horizon, S, False, 3, True, ro._speed_dt)
        scene_pass()
        torch.cuda.synchronize()
        g = graphed(scene_pass, 5, torch)
        times[key] = time_kernel(g.replay, 3, torch, min_s=0.05) / 5
        out[key] = {"us": round(times[key], 2), "cost_sum": float(ro.scene_dist.sum())}
        if key == "mesh_launch":
            scene_cells = scene
    alg = B * H * kin.num_spheres * 36
    out["mesh_launch"].update({"algorithmic_bytes": alg, "GBps": round(alg / times["mesh_launch"] * 1e-3, 1),
                               "hbm_frac": round(alg / times["mesh_launch"] * 1e-3 / HBM_PEAK_GBS, 4),
                               "cost_relative_to_cuboids": round(out["mesh_launch"]["cost_sum"] / max(out["cuboid_kernel"]["cost_sum"], 1e-9), 6),
                               "slowdown_vs_cuboid_kernel": round(times["mesh_launch"] / times["cuboid_kernel"], 2)})
    out["mesh_launch"]["kernels"] = ("curobo_hip_sphere_mesh_collision_ws: sphere_mesh_select_kernel<3> (bounding-box reject, queue) + "
                                     "sphere_mesh_cells_kernel<3> (distance-sorted closest-triangle cell lists) + sphere_mesh_wide_kernel<3> (a workgroup per sphere about "
                                     "equally far from very many triangles) + sphere_mesh_walk_kernel<3> (what the lists do not apply to); the queue counters are "
                                     "cleared by mesh_queue_reset_kernel")
    out["mesh_launch_tree_walk"]["note"] = "the same launch over meshes built without cell lists (round 4-5 form): select + tree walk"
    out["mesh_launch"]["speedup_vs_tree_walk"] = round(times["mesh_launch_tree_walk"] / times["mesh_launch"], 2)
    out["mesh_launch"]["cost_equals_tree_walk"] = bool(abs(out["mesh_launch"]["cost_sum"] - out["mesh_launch_tree_walk"]["cost_sum"])
                                                       <= 1e-5 * abs(out["mesh_launch_tree_walk"]["cost_sum"]))
    out["mesh_launch"]["cell_lists"] = [m.cells_info for m in scene_cells.meshes.meshes] if scene_cells is not None else None
    out["mesh_launch"]["kernel_counters"] = {
        k2: (lambda e: None if e is None else {"mean_us": e.get("mean_us"), "hbm_bytes": e.get("hbm_bytes"), "valu_issue_frac": e.get("valu_issue_frac"),
                                               "SQ_INSTS_VALU": e["counters"].get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": e["counters"].get("SQ_INSTS_SALU"),
                                               "source": e["source"]})(committed_counters(k2))
        for k2 in ("sphere_mesh_select_kernel", "sphere_mesh_cells_kernel", "sphere_mesh_walk_kernel")}
    # BVH build and the two bakes of one mesh (torus-free: the table box, 3 072 triangles) into a 96^3 grid
    v, f = meshes[0][0]["vertices"], meshes[0][0]["faces"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bvh = build_mesh_bvh(v, f, device, cells=False)
    torch.cuda.synchronize()
    out["bvh_build"] = {"triangles": int(len(f)), "ms": round((time.perf_counter() - t0) / 5 * 1e3, 3),
                        "note": "Morton keys + torch.sort + one launch per tree level; host-driven, per loaded mesh, once"}
    from curobo_amd.backends.mesh import build_mesh_cells
    t0 = time.perf_counter()
    for _ in range(3):
        build_mesh_cells(bvh)
    torch.cuda.synchronize()
    out["cell_lists_build"] = {"ms": round((time.perf_counter() - t0) / 3 * 1e3, 3), **(bvh.cells_info or {}),
                               "note": "count launch + prefix sum + fill launch + torch.sort of the entries; per loaded mesh, once"}
    bvh = build_mesh_bvh(v, f, device, cells=False)
    n = 96
    grid_a = torch.zeros(n ** 3, dtype=torch.float16, device=device)
    grid_b = torch.zeros_like(grid_a)
    xf = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    vt, ft = torch.as_tensor(v, device=device), torch.as_tensor(f, device=device)
    us_all = time_kernel(lambda: Cn.mesh_esdf_bake(grid_a, vt, ft, (n, n, n), 0.05, xf, max_distance=1.0), 3, torch, min_s=0.05)
    us_bvh = time_kernel(lambda: mesh_esdf_bake_bvh(grid_b, bvh, (n, n, n), 0.05, xf, max_distance=1.0), 3, torch, min_s=0.05)
    out["esdf_bake_96^3"] = {"all_triangles_us": round(us_all, 1), "through_bvh_us": round(us_bvh, 1), "speedup": round(us_all / us_bvh, 1),
                             "max_abs_difference": float((grid_a.float() - grid_b.float()).abs().max())}
    return out


def c3_benchmark(device, torch):
    """C3: UR10e + one nvblox-style ESDF grid (128^3 fp16 at 0.02 m), 512 seeds.  (a) the
    collision_checking SDF path (RobotCollisionChecker.get_scene_self_collision_distance_from_joints:
    FK -> sphere-voxel distance + self collision) on 512 seeds x 33 points; (b) the trajopt rollout
    (cost + gradient) of 512 seeds x 4 candidates on the same world, fused launch and kernel sequence."""
    from curobo_amd.collision_checking import RobotCollisionChecker
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c3_voxel_world, seed_knots, start_configuration

    kcfg = KinematicsCfg.from_packaged("ur10e", device=device)
    model, kin = kcfg.model, kcfg.kinematics_config
    arrays = c3_voxel_world()
    scene = SceneData.from_arrays(arrays, device)
    seeds, H = 512, 33
    D, S, L, T = kin.num_dof, kin.num_spheres, kin.num_links, kin.num_pose_links
    knots = seed_knots(model, seeds * 4, 12, seed=4)
    res = {"workload": "C3: UR10e, 128^3 fp16 ESDF (2.56 m cube, box + sphere union), 512 seeds",
           "robot": {"dof": D, "spheres": S, "links": L}}
    # (a) collision checker query
    chk = RobotCollisionChecker(kcfg, scene, activation_distance=0.02, scene_weight=1.0, self_weight=1.0)
    g = torch.Generator().manual_seed(3)
    lo, hi = kin.joint_limits_position[0].cpu(), kin.joint_limits_position[1].cpu()
    q = (lo + (hi - lo) * torch.rand(seeds, H, D, generator=g)).to(device)
    d_world, d_self = chk.get_scene_self_collision_distance_from_joints(q)
    torch.cuda.synchronize()
    res["hit_fraction"] = round(float((d_world > 0).float().mean()), 4)
    query = lambda: chk.get_scene_self_collision_distance_from_joints(q)  # noqa: E731
    try:
        with torch.no_grad():
            gq = graphed(query, 5, torch)
        us, how = time_kernel(gq.replay, 3, torch, min_s=0.05) / 5, "hipGraph replay"
    except Exception:  # noqa: BLE001  (front end not capturable: eager launches, Python overhead included)
        torch.cuda.synchronize()
        with torch.no_grad():
            us, how = time_kernel(query, 20, torch, min_s=0.05), "eager launches (Python overhead included)"
    n = seeds * H
    nbytes = n * (4 * D + 28 * T + 16 * S + 48 * L + 2 * 16 * S + 4 * S + 4 + 16 * S)  # FK out, 2 reads of spheres, outputs, 8 fp16 corners
    res["collision_checking_query"] = {"points": n, "us": round(us, 2), "sphere_queries_per_s": round(n * S / us * 1e6, 1),
                                       "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / us * 1e-3, 1), "timing": how}
    # (b) rollout cost + gradient
    B = seeds * 4
    x = torch.as_tensor(knots, device=device).reshape(B, -1)
    for name, fused in (("fused", True), ("kernel_sequence", False)):
        ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_fused=fused))
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        if fused and not ro.fused_available():
            res[name] = {"error": "one trajectory does not fit in LDS"}
            continue
        gr = graphed(lambda: ro.cost_and_gradient(x), 5, torch)
        us = time_kernel(gr.replay, 3, torch, min_s=0.05) / 5
        nb = B * H * (ro.algorithmic_bytes_per_point() + 16 * S)
        res[name] = {"us_per_launch_set": round(us, 2), "rollouts_per_s": round(B / us * 1e6, 1),
                     "algorithmic_bytes": int(nb), "GBps": round(nb / us * 1e-3, 1)}
        if not fused:
            N = B * H
            stages = {
                "fk_forward_spheres": (lambda: ro.compute_kinematics(ro.position), N * (4 * D + 28 * T + 16 * S + 48 * L)),
                "self_collision": (lambda: rollout_self(ro), N * (16 * S + 4)),
                "scene_collision_voxel_swept": (lambda: rollout_scene(ro), N * (36 * S + 16 * S)),
                "fk_backward": (lambda: rollout_bwd_fk(ro), N * (48 * L + 16 * S + 28 * T + 4 * D)),
            }
            res["kernels"] = _timed_stages(stages, torch)
    best = res["fused"] if "rollouts_per_s" in res.get("fused", {}) else res["kernel_sequence"]
    res["value"], res["unit"] = best["rollouts_per_s"], "rollouts/s"
    k = res["kernels"]["scene_collision_voxel_swept"]
    cr = counter_roofline("scene_collision_packed_kernel<3, 2>", 0, k["us"], units=B * H)
    res["roofline"] = {"bound": "hbm", "kernel": "scene_collision_packed_kernel (fp16 ESDF, swept)", "achieved": k["GBps"], "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": k["hbm_frac"], "traffic": cr.get("traffic"), "avg_launch_us": k["us"],
                       "algorithmic_bytes_per_launch": k["algorithmic_bytes"],
                       "note": "36*S B/pt of sphere reads / distance + gradient writes + 8 fp16 corner gathers (16 B) per sphere; "
                               "the grid (4 MiB) is L2 / Infinity-Cache resident",
                       **({"primary_bound": cr["valu"], "counters_source": cr["counters_source"], "l2_hit_rate": cr.get("l2_hit_rate")}
                          if "valu" in cr else {})}
    res["kernel_counters"] = _stage_counters(res["kernels"], {
        "fk_forward_spheres": "fk_forward_points_kernel", "self_collision": "self_collision_row16_kernel",
        "scene_collision_voxel_swept": "scene_collision_packed_kernel", "fk_backward": "fk_backward_kernel"}, B * H, hint=2048)
    return res


def _stage_counters(kernels: dict, names: dict, units: int, hint: int = 0) -> dict:
    """committed counters next to the live stage timings: {stage: {traffic, valu frac, ...}} (launch shapes are matched by
    the number of trajectories ``hint`` where several shapes of a kernel were profiled: the closest algorithmic size)"""
    out = {}
    for stage, kname in names.items():
        if stage not in kernels:
            continue
        import glob

        rows = []
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters_by_kernel.json"))):
            try:
                rec = json.load(open(path))
            except (OSError, ValueError):
                continue
            got = [dict(e, source="profiles/" + os.path.basename(path)) for e in rec.get("kernels", []) if kname in e["kernel"] and e.get("counters")]
            rows = got or rows
        if not rows:
            continue
        alg = kernels[stage]["algorithmic_bytes"]
        # the shape whose HBM traffic is closest to this stage's algorithmic bytes is the one profiled on this workload
        e = min(rows, key=lambda r: abs((r.get("hbm_bytes") or 0) - alg))
        us = kernels[stage]["us"]
        v = e["counters"].get("SQ_INSTS_VALU")
        out[stage] = {"kernel": e["kernel"], "workgroups": e["workgroups"], "traffic": e.get("hbm_bytes"),
                      "traffic_over_algorithmic": round((e.get("hbm_bytes") or 0) / alg, 3) if alg else None,
                      "hbm_frac_measured_traffic": round((e.get("hbm_bytes") or 0) / us * 1e-3 / HBM_PEAK_GBS, 4),
                      "valu_issue_frac": round(v / (us * 1e-6) / VALU_ISSUE_PEAK, 4) if v else None,
                      "l2_hit_rate": e.get("l2_hit_rate"), "source": e["source"]}
    return out


def c4_benchmark(device, torch):
    """C4 single-GPU share: Unitree G1 whole body (in-tree stand-in for the 38-DoF humanoid; 43 actuated
    dof + virtual base), 256 seeds (1024 seeds over 4 GPUs) x 4 candidates x 33 points: FK, tiled
    self collision over 162 k sphere pairs, RNEA inverse dynamics + torque-limit cost and the VJPs,
    run as the kernel sequence (one G1 trajectory does not fit in 160 KB of LDS)."""
    from curobo_amd.kinematics import KinematicsCfg
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.workloads import seed_knots, start_configuration

    kcfg = KinematicsCfg.from_packaged("unitree_g1", device=device)
    model, kin = kcfg.model, kcfg.kinematics_config
    seeds, nls, H = 256, 4, 33
    B = seeds * nls
    D, S, L, T = kin.num_dof, kin.num_spheres, kin.num_links, kin.num_pose_links
    P = int(kin.self_collision.collision_pairs.shape[0])
    eff = [200.0] * D
    cfg = TrajOptRolloutCfg(use_fused=False, use_torque_limits=True, effort_limit=eff)
    ro = TrajOptRollout(kin, None, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
    x = torch.as_tensor(seed_knots(model, B, 12, seed=6, spread=0.15), device=device).reshape(B, -1)
    ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = graphed(lambda: ro.cost_and_gradient(x), 2, torch)
    us = time_kernel(g.replay, 2, torch, min_s=0.1) / 2
    N = B * H
    res = {"workload": f"C4 share: unitree_g1 ({D} dof, {L} links, {S} spheres, {P} self-collision pairs), {seeds} seeds x {nls} "
                       f"candidates x {H} points: pose + c-space STATE (torque limits on RNEA tau) + self collision, cost+grad, "
                       "kernel sequence",
           "us_per_rollout_set": round(us, 1), "value": round(B / us * 1e6, 1), "unit": "rollouts/s",
           "in_self_collision_fraction": round(float((ro.self_dist > 0).float().mean()), 4)}
    from curobo_amd.backends import dynamics as Dy
    from curobo_amd.backends import geometry as G

    sc = kin.self_collision

    def self_only():
        G.self_collision_distance(ro.self_dist, ro.self_grad, ro._pd, ro.self_sparse, ro.robot_spheres, sc.sphere_padding,
                                  ro._w_self, sc.collision_pairs, ro._bbmv, ro._bbmi, 1, 256, B, H, S, P, False, True)
    n = N
    rargs = (kin.fixed_transforms, kin.link_masses_com, kin.link_inertias, kin.joint_map_type, kin.joint_map, kin.link_map,
             kin.joint_offset_map, ro._gravity, kin.link_level_offsets, kin.link_level_data)

    def rnea_f():
        Dy.launch_rnea_forward(ro._tau, ro.position.view(n, D), ro.velocity.view(n, D), ro.acceleration.view(n, D), *rargs,
                               ro._rnea_cache, n, L, D, kin.n_tree_levels, 1, None)

    def rnea_b():
        Dy.launch_rnea_backward(*ro._rnea_g, ro._cs_gtau.view(n, D), ro.position.view(n, D), ro.velocity.view(n, D), *rargs,
                                ro._rnea_cache, n, L, D, kin.n_tree_levels, 1, None, ro._rnea_ws)
    stages = {
        "fk_forward_spheres": (lambda: _fk_fwd(ro, kin, B, H, D, S), N * (4 * D + 28 * T + 16 * S + 48 * L)),
        "self_collision_tiled": (self_only, N * (16 * S + 4)),
        "rnea_forward": (rnea_f, N * (12 * D + 4 * D + 80 * L)),
        "rnea_backward": (rnea_b, N * (16 * D + 12 * D + 80 * L + 2 * 72 * L)),
        "fk_backward": (lambda: _fk_bwd(ro, kin, B, H, D, S), N * (48 * L + 16 * S + 28 * T + 4 * D)),
    }
    res["kernels"] = _timed_stages(stages, torch, reps=2)
    res["kernels_note"] = ("every kernel timed ALONE (plain launches); in the rollout set the joint-space chain (RNEA -> c-space -> RNEA VJP) "
                           "runs on a side stream next to the task-space chain, and its RNEA launches read their inputs from a transposed "
                           "scratch instead of LDS (curobo_hip_launch_rnea_*_scratch: 177 / 325 us alone, but the self-collision kernel "
                           "keeps its eight points per CU next to them)")
    k = res["kernels"]["self_collision_tiled"]
    cr = counter_roofline("self_collision_tiles", 0, k["us"], units=N)
    res["roofline"] = {"bound": "hbm", "kernel": "self_collision_tiles2_kernel (pair bitmap, two-level broad phase: 16- and 4-sphere boxes, 4 waves per point, 162 k pairs)",
                       "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["hbm_frac"], "traffic": cr.get("traffic"),
                       "avg_launch_us": k["us"], "algorithmic_bytes_per_launch": k["algorithmic_bytes"],
                       "listed_pair_tests_per_s": round(P * N / k["us"] * 1e6, 1),
                       "note": "the pair tests are compute (31 FLOP/B against the materialised sphere tensor): the binding bound is "
                               "VALU issue, priced with the EXECUTED instruction count of the launch (committed SQ counters of this "
                               "workload x the live launch time) -- not with the listed pairs, 6/7 of which the broad phase never "
                               "evaluates",
                       **({"primary_bound": cr["valu"], "counters_source": cr["counters_source"], "l2_hit_rate": cr.get("l2_hit_rate")}
                          if "valu" in cr else {})}
    res["kernel_counters"] = _stage_counters(res["kernels"], {
        "fk_forward_spheres": "fk_forward_kernel", "self_collision_tiled": "self_collision_tiles", "rnea_forward": "rnea_forward",
        "rnea_backward": "rnea_backward", "fk_backward": "fk_backward_kernel"}, N)
    # the whole rollout set against the same roofline: the algorithmic bytes of its five dominant stages (the per-point
    # figures above) / the measured time of one set (both chains, as the rollout runs them)
    set_bytes = int(sum(v["algorithmic_bytes"] for v in res["kernels"].values()))
    res["roofline_whole_set"] = {"bound": "hbm", "algorithmic_bytes_per_set": set_bytes, "us_per_rollout_set": res["us_per_rollout_set"],
                                 "achieved": round(set_bytes / us * 1e-3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(set_bytes / us * 1e-3 / HBM_PEAK_GBS, 4),
                                 "stages": {k2: v["algorithmic_bytes"] for k2, v in res["kernels"].items()},
                                 "note": "sum of the stages' algorithmic bytes (SURVEY 8d per-point figures x 33 792 points) over the time "
                                         "of one rollout set; the small stages (B-spline, tool pose, c-space STATE, aggregates) add time but "
                                         "no bytes here"}
    return res


def _fk_fwd(ro, kin, B, H, D, S):
    from curobo_amd.backends import kinematics as K

    K.launch_kinematics_forward_spheres(ro.link_pos, ro.link_quat, ro.robot_spheres, ro.com, ro.cumul_mat, ro.position,
                                        kin.fixed_transforms, kin.link_spheres, kin.link_masses_com, kin.joint_map_type, kin.joint_map,
                                        kin.link_map, kin.tool_frame_map, kin.link_sphere_idx_map, kin.joint_offset_map, ro.env_query_idx,
                                        kin.num_envs, B * H, H, D, S, 32, True, False)


def _fk_bwd(ro, kin, B, H, D, S):
    from curobo_amd.backends import kinematics as K

    K.launch_kinematics_backward(ro.grad_q, ro.pose_grad_pos, ro.pose_grad_quat, ro.self_grad, ro.com, ro.com, ro.pose_grad_pos,
                                 ro.cumul_mat, kin.link_spheres, kin.link_masses_com, kin.link_map, kin.joint_map, kin.joint_map_type,
                                 kin.tool_frame_map, kin.link_sphere_idx_map, kin.link_chain_data, kin.link_chain_offsets,
                                 kin.joint_links_data, kin.joint_links_offsets, kin.joint_affects_endeffector, kin.joint_offset_map,
                                 ro.env_query_idx, kin.num_envs, B * H, H, D, S, False, False)


def c5_benchmark(model, kin, device, torch):
    """C5 single-GPU share: 2 of the 16 planning problems (16 robots over 8 GPUs), each with its own
    world (cuboids + one 64^3 ESDF grid per environment, env_query_idx per trajectory), 512 seeds x 4
    candidates x 64-step horizon (12 knots x 4, padded 65): cost + gradient of 4096 trajectories."""
    from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg
    from curobo_amd.scene import SceneData
    from curobo_amd.workloads import c5_mixed_worlds, seed_knots, start_configuration

    n_prob, seeds, nls = 2, 512, 4
    B = n_prob * seeds * nls
    res = {"workload": f"C5 share: {n_prob} problems (own worlds) x {seeds} seeds x {nls} candidates, Franka, horizon 64 (padded 65), "
                       "swept scene collision + speed metric + self collision, cost+grad"}
    env_idx = torch.arange(n_prob, dtype=torch.int32, device=device).repeat_interleave(seeds * nls)
    x = torch.as_tensor(seed_knots(model, B, 12, seed=8), device=device).reshape(B, -1)
    for world_name in ("mixed cuboid + ESDF", "cuboid-only"):
        arrays = c5_mixed_worlds(n_prob, voxels=world_name.startswith("mixed"))
        scene = SceneData.from_arrays(arrays, device)
        sub = {}
        for name, fused in (("fused", True), ("kernel_sequence", False)):
            ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(interpolation_steps=4, use_fused=fused))
            ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
            ro.update_env_query_idx(env_idx)
            if fused and not ro.fused_available():
                sub[name] = {"error": "one trajectory does not fit in LDS"}
                continue
            g = graphed(lambda: ro.cost_and_gradient(x), 3, torch)
            us = time_kernel(g.replay, 3, torch, min_s=0.05) / 3
            nb = B * ro.cfg.padded_horizon * ro.algorithmic_bytes_per_point()
            sub[name] = {"us_per_launch_set": round(us, 1), "rollouts_per_s": round(B / us * 1e6, 1), "algorithmic_bytes": int(nb),
                         "GBps": round(nb / us * 1e-3, 1), "hbm_frac": round(nb / us * 1e-3 / HBM_PEAK_GBS, 4)}
        res[world_name] = sub
    head = res["mixed cuboid + ESDF"]
    best = max((v for v in head.values() if "rollouts_per_s" in v), key=lambda v: v["rollouts_per_s"])
    res["value"], res["unit"] = best["rollouts_per_s"], "rollouts/s"
    is_fused = best is head.get("fused")
    cr = counter_roofline("rollout_trajectory_fused_kernel<3, 3, 3", B, best["us_per_launch_set"], units=B) if is_fused else {}
    res["roofline"] = {"bound": "hbm", "kernel": "rollout_trajectory_fused_kernel<3, 3, 3> (H = 65, multi-env, cuboids + ESDF)" if is_fused else "kernel sequence",
                       "achieved": best["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": best["hbm_frac"], "traffic": cr.get("traffic"),
                       "avg_launch_us": best["us_per_launch_set"], "algorithmic_bytes_per_launch": best["algorithmic_bytes"],
                       **({"primary_bound": cr["valu"], "counters_source": cr["counters_source"], "l2_hit_rate": cr.get("l2_hit_rate"),
                           "note": "the fused launch keeps its intermediates in LDS: the HBM fraction is notional, VALU issue binds"}
                          if "valu" in cr else {})}
    return res


def full_trajopt_benchmark(seeds, model, kin, scene, device, torch):
    """Secondary: the FULL reference trajopt cost set (tool pose + c-space STATE + self + swept
    scene collision, lbfgs_bspline_trajopt.yml) on the C2 shapes: one fused launch vs the
    ten-launch kernel sequence, cost + gradient of 1024 trajectories."""
    from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
    from curobo_amd.workloads import seed_knots, start_configuration

    B = seeds * 4
    res = {}
    knots = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=device)
    for name, fused in (("fused_us", True), ("kernel_sequence_us", False)):
        ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(use_fused=fused))
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        x = knots.reshape(B, -1)
        g = graphed(lambda: ro.cost_and_gradient(x), 10, torch)
        res[name] = round(time_kernel(g.replay, 20, torch) / 10, 1)
    res["rollouts_per_s_fused"] = round(B / res["fused_us"] * 1e6, 1)
    res["workload"] = "C2 shapes, full trajopt cost set (pose + c-space state + self + swept scene), cost+grad"
    # the same with joint-torque limits (inverse dynamics + its VJP per point: inside the fused launch / two more launches)
    tq = {}
    for name, fused in (("fused_us", True), ("kernel_sequence_us", False)):
        cfg = TrajOptRolloutCfg(use_fused=fused)
        cfg.use_torque_limits = True
        cfg.fused_torque_max_batch = 1 << 30 if fused else 0  # (measure both forms at this batch; the rollout's own choice is below)
        ro = TrajOptRollout(kin, scene, B, cfg)
        ro.update_start_state(torch.as_tensor(start_configuration(model), device=device))
        x = knots.reshape(B, -1)
        g = graphed(lambda: ro.cost_and_gradient(x), 10, torch)
        tq[name] = round(time_kernel(g.replay, 20, torch) / 10, 1)
    tq["rollout_picks"] = "kernel_sequence" if B > TrajOptRolloutCfg().fused_torque_max_batch else "fused"
    res["with_torque_limits"] = tq
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
        # the reference's OPTIONAL convergence exit (LBFGSOptCfg.fixed_iters = False, converged_ratio 0.8; its task configs ship
        # fixed_iters = true, so this is a second reading, not the headline of this object): the optimisers of the trajopt
        # and of the IK stage stop between 25-iteration blocks once 80 % of their problems stopped improving
        cfg2 = TrajOptSolverCfg(num_seeds=S)
        cfg2.optimizer.fixed_iters = False
        cfg2.ik.optimizer.fixed_iters = False
        slv2 = TrajOptSolver(kin, scene, P, cfg2)
        r2 = slv2.solve_pose(start, gp, gq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its = []
        for _ in range(reps):
            r2 = slv2.solve_pose(start, gp, gq)
            its.append(int(getattr(slv2.optimizer, "iterations_run", -1)))
        torch.cuda.synchronize()
        res[f"{P}_problems_x_{S}_seeds"]["with_convergence_exit"] = {
            "ms_per_batch": round((time.perf_counter() - t0) / reps * 1e3, 2), "success_rate": round(float(r2.success.float().mean()), 3),
            "iterations_of_the_last_pass": its}
    res["workload"] = ("Franka, C2 world, 32-step horizon, pose goal; IK (64 seeds: Levenberg-Marquardt seed stage, its L-BFGS stage only "
                       "when the seed stage leaves a problem unsolved, exit_early as the reference's planner) + trajopt (100 iterations, "
                       "pose + c-space state + self + swept scene collision) + finetune pass + metrics; context only: the reference "
                       "publishes 31 ms mean solve time for its full motion planner on an RTX 6000 Ada")
    return res


IK_PROTOCOL_PUBLISHED_MS = {"franka": (2.601, 2.726), "dual_ur10e": (6.058, 15.64), "unitree_g1": (31.39, 526.9)}
IK_PROTOCOL_PUBLISHED_SUCCESS = {"franka": (100.0, 100.0), "dual_ur10e": (100.0, 99.2), "unitree_g1": (100.0, 98.4)}


def ik_protocol_case(robot: str, collision_free: bool, torch, batch: int = 100) -> dict:
    """One row of the reference's IK benchmark (benchmark/ik_benchmark.py:53-165; published: docs/reference/benchmarks.rst:62-72)
    over this package's ``InverseKinematics`` front end: batch 100, goals = FK of collision-free samples, `IK` = no collision
    terms with 2 seeds, `collision-free IK` = self collision + collision_table.yml with 8 (Franka) / 16 seeds, exit_early on, three
    warm-up solves, then the mean over five goal sets of the wall time of ``solve_pose`` (host clock around a synchronised call).
    The `IK` rows load the robot as the reference's benchmark does there (ik_benchmark.py:60-65: collision_link_names = None,
    lock_joints = None -> the packaged ``<robot>_kinematics_only`` models: no spheres, only the joints on the chains to the tool
    frames; goals from unrestricted joint samples); seed-solver seed counts and the G1's 240 L-BFGS iterations are the reference's."""
    import numpy as np

    from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
    from curobo_amd.types import JointState

    g1 = robot == "unitree_g1"
    seeds = (16 if robot in ("unitree_g1", "dual_ur10e") else 8) if collision_free else 2
    ik = InverseKinematics(InverseKinematicsCfg.create(
        robot=f"{robot}.yml" if collision_free else f"{robot}_kinematics_only.yml", scene_model="collision_table.yml" if collision_free else None,
        num_seeds=seeds, position_tolerance=0.005,
        optimizer_collision_activation_distance=0.0025, self_collision_check=collision_free, use_cuda_graph=True,
        seed_solver_num_seeds=128 if g1 else max(32, 2 * seeds), max_batch_size=batch, override_iters_for_multi_link_ik=240 if g1 else None))
    torch.manual_seed(2)
    sets, ratio = [], 10
    for _ in range(5):
        q = ik.sample_configs(batch, rejection_ratio=ratio)
        while q.shape[0] < batch:
            ratio = int(1.2 * ratio) + 1
            if ratio > 400:
                raise RuntimeError("rejection ratio too high")
            q = ik.sample_configs(batch, rejection_ratio=ratio)
        sets.append(q[:batch].contiguous())
    goal = lambda q: ik.compute_kinematics(JointState.from_position(q)).tool_poses.as_goal()  # noqa: E731
    ik.config.exit_early = False
    for _ in range(3):
        ik.reset_seed()
        ik.solve_pose(goal(sets[0]))
    ik.config.exit_early = True
    times, succ, perr, rerr = [], [], [], []
    for q in sets:
        ik.reset_seed()
        g = goal(q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ik.solve_pose(g)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        ok = r.success.view(-1)
        succ.append(100.0 * float(ok.float().mean()))
        if bool(ok.any()):
            perr.append(float(np.percentile(r.position_error.view(-1)[ok].cpu().numpy(), 90)))
            rerr.append(float(np.percentile(r.rotation_error.view(-1)[ok].cpu().numpy(), 90)))
    col = 1 if collision_free else 0
    return {"robot": robot, "collision_free": collision_free, "batch": batch, "num_seeds": seeds, "tool_frames": len(ik.tool_frames), "dof": ik.dof,
            "robot_model": f"{robot}.yml" if collision_free else f"{robot}_kinematics_only (collision links and locked joints stripped: ik_benchmark.py:60-65)",
            "ms": round(1e3 * float(np.mean(times)), 3), "ms_each": [round(1e3 * t, 3) for t in times], "success_percent": round(float(np.mean(succ)), 2),
            "position_error_p90_mm": round(1e3 * float(np.mean(perr)), 5) if perr else None,
            "rotation_error_p90_deg": round(float(np.degrees(np.mean(rerr))), 5) if rerr else None,
            "published_ms_nvidia": IK_PROTOCOL_PUBLISHED_MS[robot][col], "published_success_percent": IK_PROTOCOL_PUBLISHED_SUCCESS[robot][col],
            "solves_per_s": round(batch / float(np.mean(times)), 1)}


def ik_protocol_benchmark(torch) -> dict:
    """the six rows of the reference's published IK table, each guarded on its own"""
    rows = []
    for robot in ("franka", "dual_ur10e", "unitree_g1"):
        for cfree in (False, True):
            try:
                rows.append(ik_protocol_case(robot, cfree, torch))
            except Exception as e:  # noqa: BLE001
                rows.append({"robot": robot, "collision_free": cfree, "error": f"{type(e).__name__}: {str(e)[:300]}"})
    return {"rows": rows, "protocol": "reference benchmark/ik_benchmark.py: batch 100, IK (2 seeds, no collision terms) / collision-free IK (8 or 16 "
                                      "seeds, self collision + collision_table.yml), mean wall time of solve_pose over five goal sets",
            "published": "docs/reference/benchmarks.rst:62-72 (NVIDIA GPU not named on the page)"}


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
    extra = {}
    try:  # the stage that does the work of such a solve: the Levenberg-Marquardt seed solver (iterate launches + ranking)
        T, G = kin.num_pose_links, 1
        gp4, gq4 = gp.to(device).view(P, T, G, 3).contiguous(), gq.to(device).view(P, T, G, 4).contiguous()
        ss = solver.seed_solver
        lm_us = time_kernel(lambda: ss.solve_batch(gp4, gq4, return_seeds=S), 5, torch, min_s=0.05)
        n_rows, iters, D = P * ss.S, ss.cfg.max_iterations, kin.num_dof
        R = 6 * T + D
        # per (problem, seed) and iteration: J^T J (R x D x D multiply-adds on the matrix cores) + J^T e + the Cholesky solve
        flops = n_rows * iters * (2.0 * R * D * D + 2.0 * R * D + D ** 3 / 3.0 + 2.0 * D * D)
        extra["roofline"] = {"bound": "mfma", "kernel": "seed_ik_solve_kernel (FK + Jacobian + J^T J on v_mfma_f32_16x16x4_f32 + Cholesky + state update)",
                             "achieved": round(flops / lm_us * 1e-6, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(flops / lm_us * 1e-6 / FP32_VECTOR_PEAK_TFLOPS, 5), "traffic": None, "avg_launch_us": round(lm_us, 1),
                             "algorithmic_flops_per_launch": int(flops), "lm_rows": n_rows, "lm_iterations": iters,
                             "note": "13 x 7 x 7 contractions: the matrix pipe is all but idle, the solve is a chain of dependent "
                                     "small steps (latency); fp32 MFMA issues at the vector rate on gfx950"}
    except Exception as e:  # noqa: BLE001
        extra["roofline"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_cpu_baseline:
        try:
            extra["cpu_baseline"] = cpu_ik_baseline(model, kin, gp, gq, solver, min(args.cpu_seconds, 8.0))
            if "value" in extra["cpu_baseline"]:
                extra["speedup_vs_cpu"] = round(P / dt / extra["cpu_baseline"]["value"], 1)
        except Exception as e:  # noqa: BLE001
            extra["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
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
        "lm_seed_solver": bool(solver.cfg.use_lm_seed), "stream_shards": shards, **extra,
        "reference_published": {"ms_per_batch": 2.726, "success_rate": 1.0, "hardware": "an NVIDIA GPU the page does not name",
                                "source": "reference docs/reference/benchmarks.rst:62-72 (franka.yml, batch 100, collision-free IK)"},
        "workload": "C1: Franka 7-DoF, 64 seeds per problem (best 64 of 128 Levenberg-Marquardt seed-IK runs, as the "
                    "reference's use_lm_seed), 4-cuboid world, pose + joint-limit + self + scene collision checks, "
                    "goals = FK of rejection-sampled collision-free configurations, exit_early as in the reference's "
                    "ik_benchmark.py (L-BFGS is skipped when the seed-IK solutions pass every check for all problems); "
                    "full_optimizer = the same solve with the 100 L-BFGS iterations (4 line-search candidates each) forced",
    }


def cpu_ik_baseline(model, kin, gp, gq, solver, budget_s):
    """The IK half of the metric on the host cores (BASELINE config 1 is the reference's CPU-runnable case; the reference has
    no CPU path of its own, so this is the port): the work a GPU solve with exit_early does -- the Levenberg-Marquardt seed
    solver of oracle/seed_ik_ref.py (the reference's seed-IK iteration restated on the oracle's FK / Jacobian / tool-pose /
    LM-step functions, C with OpenMP underneath) over the same number of LM runs per problem, then the feasibility checks
    (self + scene collision of the solutions' spheres, limits, pose thresholds) -- on a bounded sample of the problems."""
    from oracle import load_native_oracle, load_oracle
    from oracle.seed_ik_ref import SeedIKRefCfg, solve

    try:
        orc = load_native_oracle()
    except Exception:  # noqa: BLE001
        orc = load_oracle()
    cores = usable_cores()
    orc.set_num_threads(cores)
    md = model.as_dict()
    n_lm = int(solver.seed_solver.S)
    T = int(kin.num_pose_links)
    gp_h, gq_h = gp.cpu().numpy().reshape(-1, T, 1, 3), gq.cpu().numpy().reshape(-1, T, 1, 4)
    lo, hi = np.asarray(md["joint_limits_position"], np.float32)
    rng = np.random.default_rng(0)
    cfg = SeedIKRefCfg(max_iterations=int(solver.seed_solver.cfg.max_iterations))
    from curobo_amd.scene import cuboid_scene_arrays
    from curobo_amd.workloads import c1_world

    arrays = cuboid_scene_arrays(c1_world())

    def one_pass(n_prob):
        seeds = (lo + (hi - lo) * rng.random((n_prob * n_lm, lo.shape[0]), dtype=np.float32)).astype(np.float32)
        idx = np.repeat(np.arange(n_prob, dtype=np.int32), n_lm)
        st = solve(orc, md, cfg, seeds, gp_h[:n_prob], gq_h[:n_prob], idx)
        q = st["joint_position"]
        fk = orc.kinematics_forward(q, md)
        sph = fk["robot_spheres"].reshape(len(q), 1, -1, 4)
        free = (orc.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"].reshape(-1) == 0) & \
            (orc.scene_collision(sph, arrays, 1.0, 0.0)["distance"].reshape(len(q), -1).sum(-1) == 0)
        ok = (st["final_success"] & free).reshape(n_prob, n_lm)
        return ok.any(1)

    n_prob = min(4, gp_h.shape[0])
    one_pass(n_prob)  # warm
    t0, solved, done = time.perf_counter(), 0, 0
    while True:
        ok = one_pass(n_prob)
        solved += int(ok.sum())
        done += n_prob
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": done / el, "unit": "IK solves/s", "cores": int(orc.num_threads()), "kind": "port", "success_rate": round(solved / done, 4),
            "sample": f"{done} problems ({n_prob} per pass) x {n_lm} Levenberg-Marquardt runs x {cfg.max_iterations} iterations + collision "
                      f"checks in {el:.1f} s on {int(orc.num_threads())} threads (oracle/seed_ik_ref.py over the C oracle, OpenMP; NumPy glue included)"}


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
                                r.env_query_idx, r.batch_size, r.cfg.padded_horizon, r.kin.num_spheres, r.use_multi_env,
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
