# per-kernel durations of the C4 (G1) kernels for each variant library: rocprofv3 --kernel-trace --stats over tools/run_kernels_once.py c4
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_g1
mkdir -p $OUT
cp $ROOT/curobo_amd/lib/libcurobo_hip.so $OUT/.orig.so
for n in "$@"; do
  cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
  rm -rf $OUT/trace_$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$n -- python $ROOT/tools/run_kernels_once.py c4 > $OUT/$n.log 2>&1 || echo "$n failed"
  f=$(find $OUT/trace_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n" >> $OUT/summary.txt
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("fk_forward_kernel","fk_backward","tiles2","rnea","cspace","tool_pose")):
        print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:110]}')
PY
  rm -rf $OUT/trace_$n
done
cp $OUT/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so; rm $OUT/.orig.so
cat $OUT/summary.txt
