#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call18; mkdir -p $O
python tools/trajopt_solve_probe.py 2>&1 | grep -v amdgpu.ids | head -60 | tee $O/probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/trajopt_solve_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT/$O
f=$(find trace -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-170
find trace -name "*.csv" -size +1M -delete
