"""Anatomy of ONE timed block of the driver's bench command (W untimed + K timed iterations as graphs + exchange + synchronise):
the graphs are captured with the fused launch's profile sequence on, so every rollout launch of the block leaves its device
wall-clock start / end (100 MHz); the host clock brackets the block as bench.py does.

    python tools/r05/block_anatomy.py [--steps 20] [--lead 5] [--shards 4] [--blocks 30]

Prints: host block time; device span first rollout start -> last rollout end; the step time inside each graph (rollout start to
rollout start per shard); the idle time at the graph boundary; host time before the first kernel and after the last."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd._lib import load  # noqa: E402
from curobo_amd.optim import LBFGSOptCfg, PipelinedLBFGS  # noqa: E402
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

a = sys.argv[1:]
opt_i = lambda k, d: int(a[a.index(k) + 1]) if k in a else d  # noqa: E731
K, LEAD, SH, NB, W = opt_i("--steps", 20), opt_i("--lead", 5), opt_i("--shards", 4), opt_i("--blocks", 30), opt_i("--warmup", 5)
dev = torch.device("cuda:0")
lib = load()
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg()
seeds, nls = 256, 4
start_t = torch.as_tensor(start_configuration(model), device=dev)


def shard_rollout(batch):
    ro = CollisionRollout(kin, scene, batch, cfg)
    ro.update_start_state(start_t)
    return ro.cost_and_gradient


opt = PipelinedLBFGS(LBFGSOptCfg(num_problems=seeds, inner_iters=25), shard_rollout, cfg.n_knots, kin.num_dof,
                     (kin.joint_limits_position[0], kin.joint_limits_position[1]), dev, n_shards=SH)
seed_t = torch.as_tensor(seed_knots(model, seeds, cfg.n_knots, seed=2), device=dev)
opt.reinitialize(seed_t)
rows = seeds // SH * nls
chunks = ([LEAD, K - LEAD] if 0 < LEAD < K else [K])
n_launch = sum(chunks) * SH
buf = torch.zeros((SH * len(chunks) + n_launch, rows, 16), dtype=torch.int64, device=dev)
lib.curobo_hip_rollout_fused_set_profile_sequence(buf.data_ptr(), buf.shape[0], rows)
graphs, first_block = [], []
used = 0
for n in chunks:
    used += SH  # make_graph runs one eager warm-up step per shard first
    first_block.append(used)
    graphs.append(opt.make_graph(n))
    used += n * SH
lib.curobo_hip_rollout_fused_set_profile_sequence(None, 0, 0)
warm = opt.make_graph(W)
res = []
for blk in range(NB):
    opt.reinitialize(seed_t)
    warm.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for g in graphs:
        g.replay()
        marks.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = buf.cpu().numpy().astype(np.float64) / 100.0
    rec = {"host_block_us": el * 1e6, "host_replay_returns_us": [m * 1e6 for m in marks]}
    spans = []
    for ci, n in enumerate(chunks):
        tb = t[first_block[ci]:first_block[ci] + n * SH]
        s, e = tb[:, :, 0].min(axis=1), tb[:, :, 4].max(axis=1)
        spans.append((s.min(), e.max(), s.reshape(n, SH), e.reshape(n, SH)))
    rec["device_span_us"] = spans[-1][1] - spans[0][0]
    rec["graph_spans_us"] = [sp[1] - sp[0] for sp in spans]
    rec["boundary_idle_us"] = [spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1)]
    # steady step inside the last graph: start-to-start per shard
    sl = spans[-1][2]
    rec["step_in_last_graph_us"] = float(np.median(np.diff(sl, axis=0))) if sl.shape[0] > 1 else None
    rec["first_iteration_of_last_graph_us"] = float(sl[1].min() - sl[0].min()) if sl.shape[0] > 1 else None
    res.append(rec)
med = lambda k: float(np.median([r[k] for r in res[3:]]))  # noqa: E731
out = {"steps": K, "chunks": chunks, "shards": SH,
       "host_block_us": round(med("host_block_us"), 1), "us_per_step": round(med("host_block_us") / K, 2),
       "device_span_us": round(med("device_span_us"), 1),
       "host_minus_device_us": round(med("host_block_us") - med("device_span_us"), 1),
       "graph_spans_us": [round(float(np.median([r["graph_spans_us"][i] for r in res[3:]])), 1) for i in range(len(chunks))],
       "boundary_idle_us": [round(float(np.median([r["boundary_idle_us"][i] for r in res[3:]])), 1) for i in range(len(chunks) - 1)],
       "host_replay_returns_us": [round(float(np.median([r["host_replay_returns_us"][i] for r in res[3:]])), 1) for i in range(len(chunks))],
       "step_in_last_graph_us": round(med("step_in_last_graph_us"), 2)}
print(json.dumps(out))
