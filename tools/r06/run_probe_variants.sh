# tools/r06/run_probe_variants.sh <variant> ...: the fallback probe + the bench's mesh leg per variant library
cd $GRAFT_REPO_ROOT
cp curobo_amd/lib/libcurobo_hip.so /tmp/libcurobo_hip_orig.so
for v in base "$@"; do
  echo "######## $v"
  [ "$v" != base ] && cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  timeout 600 python tools/r06/mesh_fallback_probe.py 2>&1 | grep "^meshes"
  timeout 600 python tools/r06/mesh_cells_probe.py 2>&1 | grep -i "launch\|us" | head -6
done
cp /tmp/libcurobo_hip_orig.so curobo_amd/lib/libcurobo_hip.so
