# final_run_g.sh: GPU suite, smoke, both bench commands (kernels unchanged since r05_f: no profile collection)
TAG=r05_g
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_line_driver.json 2> $OUT/${TAG}_bench_driver.err; cp bench_full.json $OUT/${TAG}_full_driver.json
timeout 300 python bench.py > $OUT/${TAG}_line_default.json 2> $OUT/${TAG}_bench_default.err; cp bench_full.json $OUT/${TAG}_full_default.json
tail -c 400 $OUT/${TAG}_line_default.json; echo; tail -c 400 $OUT/${TAG}_line_driver.json; echo
