# hardware counters of the mesh kernels of tools/r06/mesh_cells_probe.py (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06d
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
name=$(echo $grp | tr ' ' '+')
rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/r06d/pmc_$name -- python tools/r06/mesh_cells_probe.py 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r06d/pmc_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "sphere_mesh" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in agg[k].items()})
PY
rm -rf gpurun_out/r06d/pmc_$name
done
