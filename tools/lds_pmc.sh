set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lds_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in lanes rows; do
  if [ $mode = rows ]; then export CUROBO_HIP_SELF_ROWS=1; fi
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/$mode -- python $ROOT/tools/run_kernels_once.py c2 > $OUT/$mode.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/$mode/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'fused' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60],r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print("$mode",k,{c:sum(x)/len(x) for c,x in v.items()})
PY
done
rm -rf $OUT/lanes $OUT/rows
