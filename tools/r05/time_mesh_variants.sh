ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_meshvar
mkdir -p $OUT
L=$ROOT/curobo_amd/lib
cp $L/libcurobo_hip.so $OUT/.orig.so
for n in base "$@"; do
  [ $n = base ] || cp $L/variants/libcurobo_hip_$n.so $L/libcurobo_hip.so
  for rep in 1 2; do
    python $ROOT/bench.py --only mesh --no-cpu-baseline --no-ik --steps 20 --warmup 5 > /dev/null 2>> $OUT/err.log
    python -c "import json; d=json.load(open('$ROOT/bench_full.json'))['mesh_world']['mesh_launch']; print('$n', d['us'], d['cost_sum'])" >> $OUT/t.txt
  done
done
cp $OUT/.orig.so $L/libcurobo_hip.so; rm $OUT/.orig.so
cat $OUT/t.txt
