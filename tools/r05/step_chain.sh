ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_chain
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ik --no-configs > $OUT/bench.log 2> $OUT/bench.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/r05/step_chain.py $f 2000 | tee $OUT/chain.txt
tail -c 600 $OUT/bench.log
# keep the trace small enough to merge back: last 40000 lines
(head -1 $f; tail -40000 $f) > $OUT/kernel_trace_tail.csv; rm -rf $OUT/trace
