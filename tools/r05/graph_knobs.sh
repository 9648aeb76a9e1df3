ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_knobs
mkdir -p $OUT
run() {
  echo -n "$1 | " >> $OUT/sweep.txt
  env $1 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ik --no-configs 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $OUT/sweep.txt
}
run "X=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=8"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=32"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=256"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=2"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=8"
run "HSA_ENABLE_INTERRUPT=0"
run "X=2"
cat $OUT/sweep.txt
