"""rocprofv3 output of tools/collect_profiles_r03.sh -> the small files kept under profiles/:

  <tag>_kernel_stats_bench_c2.csv / _driver_cmd.csv   rocprofv3's own --stats table of the two bench commands
  <tag>_kernel_by_shape.csv                            per (kernel, workgroups) of the default command: calls, mean / min / max us
  <tag>_counters_by_kernel.json / .csv                 per (kernel, workgroups) of tools/run_kernels_once.py: launches, mean
                                                       duration (kernel trace WITHOUT counters), per-launch counter means and
         hbm_bytes      = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024   (gfx950: 128-B reads are tallied at 64 B,
                          /opt/skills/guides/MI355X_MICROARCH.md, HBM section; WRITE_SIZE as reported)
         l2_hit_rate    = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
         valu_issue_frac = SQ_INSTS_VALU / (duration x 256 CU x 4 SIMD x 2.4 GHz / 2 cycles)   (the guide's v_fma_f32 row)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0


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


def shape_of(r):
    grid, wg = col(r, "Grid_Size_X", "Grid_Size"), col(r, "Workgroup_Size_X", "Workgroup_Size")
    return int(grid or 0) // max(int(wg or 1), 1)


def durations(out, sub):
    d = defaultdict(list)
    for r in rows(os.path.join(out, sub, "**", "*kernel_trace.csv")):
        name = col(r, "Kernel_Name", "Name")
        t0, t1 = col(r, "Start_Timestamp", "BeginNs"), col(r, "End_Timestamp", "EndNs")
        if name is None or t0 is None or "curobo_hip" not in name:
            continue
        d[(short(name), shape_of(r))].append((int(t1) - int(t0)) / 1e3)
    return d


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summ = os.path.join(out, "summary")
    os.makedirs(summ, exist_ok=True)
    for sub, name in (("trace_default", "bench_c2"), ("trace_driver", "bench_driver_cmd"), ("trace_kernels", "run_kernels_once")):
        for path in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(summ, f"{tag}_kernel_stats_{name}.csv"))
    for sub, name in (("trace_default", "kernel_by_shape"), ("trace_driver", "kernel_by_shape_driver_cmd")):
        d = durations(out, sub)
        if d:
            with open(os.path.join(summ, f"{tag}_{name}.csv"), "w") as fh:
                fh.write("kernel,workgroups,calls,mean_us,min_us,max_us\n")
                for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
                    fh.write(f"\"{k}\",{g},{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f}\n")
    dur = durations(out, "trace_kernels")
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in glob.glob(os.path.join(out, "pmc_*")):
        if not os.path.isdir(sub):
            continue
        for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
            name = col(r, "Kernel_Name", "Name")
            if name is None or "curobo_hip" not in name:
                continue
            ctr[(short(name), shape_of(r))][col(r, "Counter_Name")].append(float(col(r, "Counter_Value")))
    rec = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python tools/run_kernels_once.py c2 c3 c4 c5 "
                      "(one counter group per run; durations from a separate --kernel-trace --stats run of the same command "
                      "without counters; tools/collect_profiles_r03.sh)",
           "corrections": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 rocprofv3 tallies 128-B read requests at 64 B "
                          "(MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported, both in KB; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                          "SQ_ACTIVE_INST_* count quad-cycles; valu_issue_frac = SQ_INSTS_VALU / (duration x 256 CU x 4 SIMD x "
                          "2.4 GHz / 2 cycles per wave64 fp32 instruction)",
           "kernels": []}
    for key in sorted(set(dur) | set(ctr), key=lambda k: (-sum(dur.get(k, [0])), k)):
        k, g = key
        c = {n: sum(v) / len(v) for n, v in ctr.get(key, {}).items()}
        d = dur.get(key, [])
        e = {"kernel": k, "workgroups": g, "launches_timed": len(d), "mean_us": round(sum(d) / len(d), 2) if d else None,
             "min_us": round(min(d), 2) if d else None, "counter_launches": len(next(iter(ctr[key].values()))) if key in ctr else 0,
             "counters": {n: round(v, 2) for n, v in sorted(c.items())}}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)
            if d:
                e["hbm_GBps"] = round(e["hbm_bytes"] / (sum(d) / len(d)) * 1e-3, 1)
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "SQ_INSTS_VALU" in c and d:
            e["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] / (sum(d) / len(d) * 1e-6) / VALU_ISSUE_PEAK, 4)
        if c.get("SQ_WAVE_CYCLES"):
            e["valu_active_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"], 4)
            e["lds_issue_stall_share_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        rec["kernels"].append(e)
    with open(os.path.join(summ, f"{tag}_counters_by_kernel.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    names = sorted({n for e in rec["kernels"] for n in e["counters"]})
    with open(os.path.join(summ, f"{tag}_counters_by_kernel.csv"), "w") as fh:
        fh.write("kernel,workgroups,launches_timed,mean_us,hbm_bytes,l2_hit_rate,valu_issue_frac," + ",".join(names) + "\n")
        for e in rec["kernels"]:
            fh.write(",".join(str(x) for x in [f"\"{e['kernel']}\"", e["workgroups"], e["launches_timed"], e["mean_us"], e.get("hbm_bytes", ""),
                                               e.get("l2_hit_rate", ""), e.get("valu_issue_frac", "")] + [e["counters"].get(n, "") for n in names]) + "\n")
    print(f"{len(rec['kernels'])} (kernel, shape) rows; with counters: {sum(1 for e in rec['kernels'] if e['counters'])}")


if __name__ == "__main__":
    main()
