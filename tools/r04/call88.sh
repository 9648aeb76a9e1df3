#!/bin/bash
# counters of the C2 kernels (recompiled after the obstacle-rotation change) and of the mesh kernels, final code
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call88; mkdir -p $O
COUNTERS_ONLY=1 timeout 900 bash tools/collect_profiles_r04.sh r04_d c2 mesh > $O/collect.log 2>&1; tail -6 $O/collect.log
