ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cp $ROOT/curobo_amd/lib/libcurobo_hip.so /tmp/.orig.so
for n in "$@"; do
  cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
  echo "== $n"; python $ROOT/tools/ik_phases.py 2>/dev/null | head -3
done
cp /tmp/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so
