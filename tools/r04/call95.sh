#!/bin/bash
# round 4: select kernel, CH spheres per lane (1 / 2 / 4 / 8 = default / 16): launch time, kernel times under the profiler, outputs bitwise
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call95; mkdir -p $O
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
export TMPDIR=/tmp
run() {
  echo "== $1"; timeout 120 python tools/r04/mesh_ab.py /tmp/$1.npz 2>&1 | grep "walk mode"
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python $GRAFT_REPO_ROOT/tools/r04/mesh_ab.py > /dev/null 2>&1)
  cat $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) | grep -i "mesh_select\|mesh_walk" | cut -d, -f1-4 | cut -c1-160
}
run default
for v in ch1 ch2 ch4 ch16; do cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so; run $v; python tools/r04/mesh_ab.py --compare /tmp/default.npz /tmp/$v.npz | grep -c identical; done
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/mesh_tests.log 2>&1; tail -3 $O/mesh_tests.log
