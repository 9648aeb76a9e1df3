ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cp $ROOT/curobo_amd/lib/libcurobo_hip.so /tmp/.orig.so
for n in "$@"; do
  cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
  python $ROOT/bench.py --no-configs --no-ik --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>/tmp/err.log
  python - <<PY
import json
d=json.load(open("$ROOT/bench_full.json"))
print("$n", d["roofline"].get("kernels_us"), "seq", d["roofline"].get("kernel_sequence_us"))
PY
done
cp /tmp/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so
