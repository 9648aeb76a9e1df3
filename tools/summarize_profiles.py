"""Condense rocprofv3 output (tools/collect_profiles.sh) into the small files kept under profiles/:
  <tag>_kernel_stats_bench_c2.csv     rocprofv3's own --stats table (kernels)
  <tag>_kernel_by_shape.csv           per (kernel, grid size): calls, mean / min / max duration
  <tag>_pmc_fused.json                HBM bytes per launch of the fused rollout kernel by launch shape
                                      (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) and its L2 hit rate
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def col(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return r[n]
    return None


def short(name):
    name = name.replace("void ", "").replace("curobo_hip::", "")
    return name.split("(")[0]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summ = os.path.join(out, "summary")
    os.makedirs(summ, exist_ok=True)
    for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(summ, f"{tag}_kernel_stats_bench_c2.csv"))
    shape = defaultdict(list)
    for r in rows(os.path.join(out, "trace", "**", "*kernel_trace.csv")):
        name = col(r, "Kernel_Name", "Name")
        t0, t1 = col(r, "Start_Timestamp", "BeginNs"), col(r, "End_Timestamp", "EndNs")
        grid, wg = col(r, "Grid_Size_X", "Grid_Size"), col(r, "Workgroup_Size_X", "Workgroup_Size")
        if name is None or t0 is None or "curobo_hip" not in name:
            continue
        shape[(short(name), int(grid or 0) // max(int(wg or 1), 1))].append((int(t1) - int(t0)) / 1e3)
    with open(os.path.join(summ, f"{tag}_kernel_by_shape.csv"), "w") as fh:
        fh.write("kernel,workgroups,calls,mean_us,min_us,max_us\n")
        for (k, g), d in sorted(shape.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f"\"{k}\",{g},{len(d)},{sum(d) / len(d):.2f},{min(d):.2f},{max(d):.2f}\n")
    # ---- counters of the fused kernel by launch shape
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in glob.glob(os.path.join(out, "pmc_*")):
        if not os.path.isdir(sub):
            continue
        for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
            name = col(r, "Kernel_Name", "Name")
            if name is None or "rollout_trajectory_fused" not in name:
                continue
            grid, wg = col(r, "Grid_Size", "Grid_Size_X"), col(r, "Workgroup_Size", "Workgroup_Size_X")
            ctr[int(grid or 0) // max(int(wg or 1), 1)][col(r, "Counter_Name")].append(float(col(r, "Counter_Value")))
    rec = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-ik "
                      "--steps 20 --warmup 10 --shards 1 --seeds <trajectories/4> (groups: FETCH_SIZE WRITE_SIZE | TCC_HIT_sum "
                      "TCC_MISS_sum, one per pass; single-stream variant of the default command, see tools/collect_profiles.sh)",
           "kernel": "rollout_trajectory_fused_kernel",
           "correction": "gfx950 rocprofv3 tallies 128-B read requests at 64 B: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE as reported; both in KB",
           "hbm_bytes_per_launch_by_trajectories": {}, "raw_by_trajectories": {}}
    for g, c in sorted(ctr.items()):
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        rec["raw_by_trajectories"][str(g)] = {**{k: round(v, 3) for k, v in mean.items()}, "dispatches": len(next(iter(c.values())))}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rec["hbm_bytes_per_launch_by_trajectories"][str(g)] = round((2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0, 1)
        if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean and mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"] > 0:
            rec["raw_by_trajectories"][str(g)]["l2_hit_rate"] = round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
    with open(os.path.join(summ, f"{tag}_pmc_fused.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec["hbm_bytes_per_launch_by_trajectories"]))


if __name__ == "__main__":
    main()
