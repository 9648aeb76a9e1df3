ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_ik
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/tools/r05/ik_timeline.py run > $OUT/run.log 2> $OUT/run.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/r05/ik_timeline.py show $f | tee $OUT/timeline.txt
grep -i "solve\|ms" $OUT/run.err | tail -3
rm -rf $OUT/trace
