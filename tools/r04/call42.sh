#!/bin/bash
# round 4, session 3: self-scheduled mesh walk with quorum transitions: bitwise vs the eight-at-a-time walk, quorum sweep, counters
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call42; mkdir -p $O
export CUROBO_MESH_WALK=1; timeout 180 python tools/r04/mesh_ab.py /tmp/walk1.npz > $O/walk1.log 2>&1; tail -2 $O/walk1.log
export CUROBO_MESH_WALK=2; timeout 180 python tools/r04/mesh_ab.py /tmp/walk2.npz > $O/walk2.log 2>&1; tail -2 $O/walk2.log
python tools/r04/mesh_ab.py --compare /tmp/walk1.npz /tmp/walk2.npz
for b in 256 4096; do for m in 1 2; do echo "== batch $b mode $m"; BATCH=$b CUROBO_MESH_WALK=$m timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"; done; done
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
for v in q2 q4 q5 q6 q8 stats; do
  cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  echo "== $v mode 2"; CUROBO_MESH_WALK=2 timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep -A1 "walk mode"
done
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
unset CUROBO_MESH_WALK
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/mesh_tests.log 2>&1; tail -3 $O/mesh_tests.log
