cd $GRAFT_REPO_ROOT
for hd in 1 0; do
echo "######## main (dual) head=$hd"; CUROBO_MESH_HEAD=$hd bash tools/r06/prof_cells.sh "BATCH=1024" "BATCH=256" 2>&1 | grep -E "cell_lists:|cells_|select|differ|rror|=="
done
cp curobo_amd/lib/libcurobo_hip.so /tmp/orig.so; cp curobo_amd/lib/variants/libcurobo_hip_nodual.so curobo_amd/lib/libcurobo_hip.so
for hd in 1 0; do
echo "######## nodual head=$hd"; CUROBO_MESH_HEAD=$hd bash tools/r06/prof_cells.sh "BATCH=1024" "BATCH=256" 2>&1 | grep -E "cell_lists:|cells_|differ|rror|=="
done
cp /tmp/orig.so curobo_amd/lib/libcurobo_hip.so
