#!/bin/bash
# tools/r05/build_one_variant.sh <name> <source stem> [flags]: ONE source with extra flags, linked with the package's other objects
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; stem=$2; shift; shift
mkdir -p $ROOT/curobo_amd/lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=fast -fno-fast-math -Wall -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I$ROOT/include -I$ROOT/curobo_amd/csrc"
[ -n "$RELINK_ONLY" ] || /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $ROOT/curobo_amd/csrc/$stem.hip -o $ROOT/curobo_amd/lib/variants/${stem}_$name.o
objs="$ROOT/curobo_amd/lib/variants/${stem}_$name.o"
for s in runtime kinematics self_collision scene_collision trajectory optimization cost dynamics rollout_fused $(cd $ROOT/curobo_amd/build && ls rollout_fused_shape*.o | sed 's/\.o$//') linalg mppi seed_ik mesh_bake mesh_bvh; do
  [ "$s" = "$stem" ] || objs="$objs $ROOT/curobo_amd/build/$s.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so $objs
echo $ROOT/curobo_amd/lib/variants/libcurobo_hip_$name.so
