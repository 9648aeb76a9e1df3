# a longer pass of the fuzzers that exercise round-6 code, new seeds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_fuzz
mkdir -p $OUT; cd $ROOT
: > $OUT/fuzz_long.txt
for f in "fuzz_mesh.py 600 701" "fuzz_mesh.py 400 702 --open" "fuzz_mesh.py 400 707 --deep" "fuzz_mesh.py 200 708 --deep --open" "fuzz_trajopt.py 300 703" "fuzz_planner.py 12 704" "fuzz_ik.py 20 705" "fuzz_opt.py 60 706"; do
  echo "== $f" >> $OUT/fuzz_long.txt
  timeout 1200 python tests/randomised/$f 2>&1 | grep -v amdgpu | tail -3 >> $OUT/fuzz_long.txt
done
cat $OUT/fuzz_long.txt
