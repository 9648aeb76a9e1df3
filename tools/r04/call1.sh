#!/bin/bash
# round 4, GPU call 1: new parity tests (measured errors), full GPU suite, counter list, I-cache counters, phase profile
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call1
mkdir -p $O
python -m pytest tests/test_gpu_dynamics.py tests/test_gpu_trajopt.py tests/test_gpu_parity_benchmarked.py -q -s -m gpu \
  -k "c4_shape or side_stream or per_sphere" > $O/parity_tests.log 2>&1
echo "parity rc=$?" >> $O/parity_tests.log
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  n=$(echo $c | tr ' ' '_')
  timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$n -- python $GRAFT_REPO_ROOT/tools/run_fused_once.py > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1 || echo "$c failed" >> $GRAFT_REPO_ROOT/$O/pmc_fail.log
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, collections, os
out="gpurun_out/r04_call1"
acc=collections.defaultdict(list)
for p in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rollout_trajectory_fused" in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
find $O -name "*.csv" -size +2M -delete
for b in 256 512 1024; do python tools/profile_fused.py --batch $b > $O/profile_fused_$b.txt 2>&1; done
tail -5 $O/parity_tests.log; tail -3 $O/gpu_suite.log; cat $O/pmc_summary.txt | head -30
