# per-kernel times of tools/r06/mesh_cells_probe.py under rocprofv3 for a list of environment settings ("BATCH=256 CAP=4096" ...)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06a
i=0
for setting in "$@"; do
i=$((i+1))
env $setting rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06a/prof_$i -- python tools/r06/mesh_cells_probe.py 10 > gpurun_out/r06a/probe_$i.txt 2>&1
echo "== $setting"
grep -E "tree_walk:|cell_lists:|identical|differ|Error|error" gpurun_out/r06a/probe_$i.txt
python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/r06a/prof_$i/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sphere_mesh" in r["Name"]: print("   ", r["Name"][:60], r["Calls"], r["AverageNs"], r.get("MinNs"), r.get("MaxNs"))
PY
rm -rf gpurun_out/r06a/prof_$i
done
