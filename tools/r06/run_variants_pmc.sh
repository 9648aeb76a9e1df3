# per variant: times (prof_cells.sh) and the instruction counters of the cell-list kernel
cd $GRAFT_REPO_ROOT
cp curobo_amd/lib/libcurobo_hip.so /tmp/libcurobo_hip_orig.so
for v in "$@"; do
  echo "######## $v"
  [ "$v" != "main" ] && cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  bash tools/r06/prof_cells.sh "BATCH=1024" "BATCH=256" 2>&1 | grep -E "cell_lists:|cells_|rror|=="
  cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/r06d/pmc_v -- python tools/r06/mesh_cells_probe.py 3 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r06d/pmc_v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "cells" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg: print("   ", k, {c: round(v / max(cnt[(k, c)], 1) / 1e6, 1) for c, v in agg[k].items()}, "M wave-instructions")
PY
  rm -rf gpurun_out/r06d/pmc_v
  cp /tmp/libcurobo_hip_orig.so curobo_amd/lib/libcurobo_hip.so
done
