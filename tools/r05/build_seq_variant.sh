#!/bin/bash
# tools/r05/build_seq_variant.sh <name> [flags]: kinematics / self_collision / scene_collision with extra flags, linked with the package's other objects
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $ROOT/curobo_amd/lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=fast -fno-fast-math -Wall -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I$ROOT/include -I$ROOT/curobo_amd/csrc"
objs=""
for s in kinematics self_collision scene_collision; do
  extra=""; [ $s = self_collision ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -x hip -c $ROOT/curobo_amd/csrc/$s.hip -o $ROOT/curobo_amd/lib/variants/${s}_$name.o &
  objs="$objs $ROOT/curobo_amd/lib/variants/${s}_$name.o"
done
wait
for s in runtime trajectory optimization cost dynamics rollout_fused rollout_fused_shape1 rollout_fused_shape2 rollout_fused_shape3 rollout_fused_shape4 linalg mppi seed_ik mesh_bake mesh_bvh; do objs="$objs $ROOT/curobo_amd/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so $objs
echo $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so
