#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r04_call10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/r04/mesh_stats.py > $O/out.txt 2>&1
cd $O
f=$(find trace -name "*kernel_stats.csv" | head -1); grep -i "mesh\|Name" $f | cut -c1-200
find trace -name "*.csv" -size +1M -delete
