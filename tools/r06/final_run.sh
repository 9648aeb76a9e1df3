# final_run.sh <tag>: the round-end sequence on the GPU box: GPU suite, smoke, both bench commands, the reference's own tests over the
# library (when .refstage/ travelled), the profile collection (kernel stats + counters incl. the ik config)
TAG=${1:-r06_c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
timeout 600 python bench.py > $OUT/${TAG}_line_default.json 2> $OUT/${TAG}_bench_default.err; cp bench_full.json $OUT/${TAG}_full_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_line_driver.json 2> $OUT/${TAG}_bench_driver.err; cp bench_full.json $OUT/${TAG}_full_driver.json
tail -c 300 $OUT/${TAG}_line_default.json; echo; tail -c 300 $OUT/${TAG}_line_driver.json; echo
if [ -d .refstage ]; then
  timeout 900 python tools/reference_on_hip.py run > $OUT/${TAG}_reference_on_hip.log 2>&1; tail -3 $OUT/${TAG}_reference_on_hip.log
  cp $OUT/ref_on_hip/report.json $OUT/${TAG}_reference_on_hip.json 2>/dev/null
fi
bash tools/collect_profiles_r06.sh $TAG > $OUT/${TAG}_collect.log 2>&1; tail -5 $OUT/${TAG}_collect.log
