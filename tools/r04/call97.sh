#!/bin/bash
# round 4: every GPU fuzzer on fresh seeds (101, 202), bounded
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call97; mkdir -p $O
for seed in 101 202; do
for job in "fuzz_fused.py 20" "fuzz_trajopt.py 12" "fuzz_scene.py 40" "fuzz_fk_bspline.py 30" "fuzz_rnea.py 6" "fuzz_mesh.py 12" "fuzz_costs.py 30" "fuzz_lm.py 40" "fuzz_mppi.py 40" "fuzz_opt.py 30" "fuzz_ik.py 4"; do
  set -- $job
  timeout 40 python tests/randomised/$1 $2 $seed > $O/${1%.py}_$seed.log 2>&1
  echo "$1 $seed rc=$? : $(grep -v amdgpu.ids $O/${1%.py}_$seed.log | grep -i "fail" | tail -3 | tr '\n' '|' | cut -c1-400)"
done; done
