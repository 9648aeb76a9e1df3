#!/bin/bash
# compiler scheduling variants of rollout_fused.hip: exclusive launch time + output checksum
cd "$GRAFT_REPO_ROOT"
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
echo "== default"; timeout 120 python tools/r04/fused_time.py 1024 256 2>&1 | grep batch
for v in $(ls curobo_amd/lib/variants | sed 's/libcurobo_hip_//; s/\.so//'); do
  cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  echo "== $v"; timeout 120 python tools/r04/fused_time.py 1024 256 2>&1 | grep batch
done
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
echo "== default again"; timeout 120 python tools/r04/fused_time.py 1024 256 2>&1 | grep batch
