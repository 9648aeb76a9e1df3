#!/bin/bash
# every randomised sweep at a larger size, three more seeds
cd "$GRAFT_REPO_ROOT"
for s in 11 12 13; do
  for f in "fuzz_fused.py 120" "fuzz_trajopt.py 60" "fuzz_scene.py 80" "fuzz_mesh.py 30" "fuzz_fk_bspline.py 60" "fuzz_rnea.py 10"; do
    set -- $f
    echo "== $1 seed $s: $(timeout 900 python tests/randomised/$1 $2 $s 2>&1 | grep -v amdgpu.ids | grep -i "failed" | cut -c1-300 | tail -3 | tr '\n' ' ')"
  done
done
timeout 600 python tests/randomised/fuzz_self.py 7 2>&1 | grep "mismatching"
