# tools/r06/mesh_world_profile.sh: kernel trace of the planner in the pillar-as-mesh world (tools/r06/mesh_world_solve_time.py, mesh rows only)
cd /tmp && export TMPDIR=/tmp
ONLY_MESH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mw -o run -- python $GRAFT_REPO_ROOT/tools/r06/mesh_world_solve_time.py > /tmp/mw.log 2>&1
grep "^mesh\|^cuboid" /tmp/mw.log
f=$(find /tmp/prof_mw -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06_mesh_world
cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_mesh_world/kernel_stats.csv
head -24 $f | cut -c1-150
