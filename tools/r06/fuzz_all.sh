ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_fuzz
mkdir -p $OUT; cd $ROOT
for f in "fuzz_fused.py 400 501" "fuzz_trajopt.py 200 502" "fuzz_scene.py 150 503" "fuzz_self.py 40 504" "fuzz_fk_bspline.py 150 505" "fuzz_rnea.py 80 506" "fuzz_mesh.py 200 607" "fuzz_mesh.py 120 611 --open" "fuzz_mesh.py 120 613 --deep" "fuzz_planner.py 6 508"; do
  echo "== $f" >> $OUT/fuzz.txt
  timeout 600 python tests/randomised/$f 2>&1 | grep -v amdgpu | tail -2 >> $OUT/fuzz.txt
done
echo "== fuzz_fused.py 40 509 with run-time shapes" >> $OUT/fuzz.txt
CUROBO_HIP_JIT_SHAPES=1 timeout 900 python tests/randomised/fuzz_fused.py 40 509 2>&1 | grep -v amdgpu | tail -2 >> $OUT/fuzz.txt
cat $OUT/fuzz.txt
