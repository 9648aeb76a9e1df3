ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_trajopt
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/tools/r05/trajopt_timeline.py run 1 8 > $OUT/run.log 2> $OUT/run.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/r05/trajopt_timeline.py show $f | tee $OUT/timeline.txt
grep solve $OUT/run.err
rm -rf $OUT/trace
