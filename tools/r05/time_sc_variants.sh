cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cp $ROOT/curobo_amd/lib/libcurobo_hip.so /tmp/.orig.so
for n in "$@"; do
  cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
  rm -rf /tmp/sc_$n; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_$n -- python $ROOT/tools/run_kernels_once.py c4 > /dev/null 2>&1
  f=$(find /tmp/sc_$n -name "*kernel_stats.csv" | head -1); echo "$n: $(grep self_collision $f | cut -d, -f1,4,6 | cut -c1-100)"
done
cp /tmp/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so
