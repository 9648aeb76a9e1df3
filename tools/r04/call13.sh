#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r04_call15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $GRAFT_REPO_ROOT/tools/r04/mesh_stats.py > $O/pmc_$n.log 2>&1 || echo "$c failed"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
out="gpurun_out/r04_call15"
acc=collections.defaultdict(list)
for p in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "sphere_mesh" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, round(sum(v)/len(v)), len(v))
PY
find $O -name "*.csv" -size +1M -delete
