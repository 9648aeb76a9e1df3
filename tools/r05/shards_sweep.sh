ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_shards
mkdir -p $OUT
for sh in 2 4 8 16; do
  for cmd in "--steps 100 --warmup 10" "--steps 20 --warmup 5"; do
    echo -n "shards $sh $cmd: " >> $OUT/sweep.txt
    python $ROOT/bench.py --gpus 1 $cmd --shards $sh --no-cpu-baseline --no-ik --no-configs 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('per_shard_launch_in_graph',{}).get('avg_launch_us'), d['roofline'].get('per_shard_launch_in_graph',{}).get('concurrent_launches'))" >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
