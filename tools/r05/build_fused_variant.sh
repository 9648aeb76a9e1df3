#!/bin/bash
# tools/r05/build_fused_variant.sh <name> [extra hipcc flags]: rollout_fused.hip with ONLY the C2 instantiation (+ flags) linked
# with the package's other objects into curobo_amd/lib/variants/libcurobo_hip_<name>.so (git-ignored; travels to the GPU box).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $ROOT/curobo_amd/lib/variants
obj=$ROOT/curobo_amd/lib/variants/rollout_fused_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=fast -fno-fast-math -Wall -Wno-unused-function \
  -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I$ROOT/include -I$ROOT/curobo_amd/csrc -DCUROBO_FUSED_ONLY_C2 "$@" \
  -x hip -c $ROOT/curobo_amd/csrc/rollout_fused.hip -o $obj
objs=""
for s in runtime kinematics self_collision scene_collision trajectory optimization cost dynamics linalg mppi seed_ik mesh_bake mesh_bvh; do objs="$objs $ROOT/curobo_amd/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so $obj $objs
echo $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so
