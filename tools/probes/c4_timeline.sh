#!/bin/bash
# kernel-trace timeline of the C4 rollout set with the joint-space chain on a side stream: start / end of every kernel of the last replay
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4tl -- python $GRAFT_REPO_ROOT/tools/c4_overlap_probe.py > /tmp/c4tl.log 2>&1
f=$(find /tmp/c4tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "curobo_hip" in r["Kernel_Name"] or "elementwise" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last occurrence of the self-collision kernel marks the last replay: print the kernels of the 1.3 ms before its end
last = [r for r in rows if "tiles2" in r["Kernel_Name"]][-1]
t_end = int(last["End_Timestamp"]) + 400_000
t0 = t_end - 1_500_000
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t_end]
base = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"].replace("void ", "").replace("curobo_hip::", "").split("(")[0][:44]
    print(f"{(int(r['Start_Timestamp']) - base) / 1e3:9.1f} {(int(r['End_Timestamp']) - base) / 1e3:9.1f}  q{r['Queue_Id']:>2}  {n}")
PY
