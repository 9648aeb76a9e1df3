# tools/r06/run_variants.sh <variant> ...: per variant library, tools/r06/prof_cells.sh with the settings in $SETTINGS
cd $GRAFT_REPO_ROOT
cp curobo_amd/lib/libcurobo_hip.so /tmp/libcurobo_hip_orig.so
for v in "$@"; do
  echo "######## $v"
  cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  bash tools/r06/prof_cells.sh "${SETTINGS:-CAP=4096 PAD=0.2}" 2>&1 | grep -E "cell_lists:|cells_|differ|rror"
done
cp /tmp/libcurobo_hip_orig.so curobo_amd/lib/libcurobo_hip.so
