#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r04_call4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --only mesh --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd $O
f=$(find trace -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220
find trace -name "*.csv" -size +1M -delete
