cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
n=$1
cp $ROOT/curobo_amd/lib/libcurobo_hip.so /tmp/.orig.so
cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d; timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $ROOT/tools/run_kernels_once.py c4 > /dev/null 2>&1 || echo "$grp failed"
  python - $d <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "self_collision" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"  {k:28s} {sum(v)/len(v):16.0f}")
PY
done
cp /tmp/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so
