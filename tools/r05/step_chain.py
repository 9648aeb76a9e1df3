"""Anatomy of one bench step from a rocprofv3 kernel trace of `bench.py --steps K` (csv): per shard stream the chain
rollout launch -> gap -> optimiser tail -> gap -> next rollout launch.  The shards of PipelinedLBFGS are independent chains of
this form; the step time is the length of one chain link, not the sum of kernel times.

    python tools/r05/step_chain.py <kernel_trace.csv> [n_last]"""
import collections
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r.get("Queue_Id"), r.get("Stream_Id"))
      for r in rows]
roll = [e for e in ev if "rollout_trajectory_fused" in e[2] and e[3] == 256]
tail = [e for e in ev if "lbfgs_iteration_tail" in e[2]]
roll, tail = roll[-n_last:], tail[-n_last:]
print(f"{len(roll)} shard rollout launches, {len(tail)} tails; queues {collections.Counter(e[4] for e in roll)}; streams {collections.Counter(e[5] for e in roll)}")
med = lambda v: round(statistics.median(v) / 1e3, 2) if v else None  # noqa: E731
print("rollout launch (256 trajectories) us: median", med([e[1] - e[0] for e in roll]), " tail us: median", med([e[1] - e[0] for e in tail]))
# chain reconstruction: a tail follows the rollout that ended last before it started ON THE SAME queue/stream key; fall back to
# nearest-in-time pairing when the trace has one queue for everything
allk = sorted(roll + tail)
key = (lambda e: e[5]) if len(set(e[5] for e in roll)) > 1 else ((lambda e: e[4]) if len(set(e[4] for e in roll)) > 1 else None)
if key is None:
    print("one queue / stream in the trace: pairing by time")
    g1, g2 = [], []
    ends = sorted(e[1] for e in roll)
    import bisect
    for t in tail:
        i = bisect.bisect_right(ends, t[0]) - 1
        if i >= 0:
            g1.append(t[0] - ends[i])
    tends = sorted(e[1] for e in tail)
    for r in roll:
        i = bisect.bisect_right(tends, r[0]) - 1
        if i >= 0:
            g2.append(r[0] - tends[i])
    print("gap rollout end -> nearest later tail start us:", med(g1), " gap tail end -> nearest later rollout start us:", med(g2))
else:
    chains = collections.defaultdict(list)
    for e in allk:
        chains[key(e)].append(e)
    g_rt, g_tr, link = [], [], []
    for k, c in chains.items():
        for a, b in zip(c, c[1:]):
            ra, rb = "rollout" in a[2], "rollout" in b[2]
            if ra and not rb:
                g_rt.append(b[0] - a[1])
            elif not ra and rb:
                g_tr.append(b[0] - a[1])
        rs = [e for e in c if "rollout" in e[2]]
        link += [b[0] - a[0] for a, b in zip(rs, rs[1:])]
    f = lambda v: [x for x in v if x < 200e3]  # noqa: E731  (drop block boundaries)
    print(f"per chain: gap rollout->tail {med(f(g_rt))} us, gap tail->rollout {med(f(g_tr))} us, chain link (rollout start -> next rollout start) {med(f(link))} us")
