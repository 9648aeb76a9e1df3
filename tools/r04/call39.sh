#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pcs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 170 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d /tmp/pcs_st -- python $GRAFT_REPO_ROOT/tools/r04/mesh_stats.py > $O/stoch.log 2>&1; echo "stochastic rc=$?"; tail -3 $O/stoch.log
find /tmp/pcs_st -type f | head; 
if ! find /tmp/pcs_st -name "*pc_sampling*" | grep -q .; then
timeout 170 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d /tmp/pcs_ht -- python $GRAFT_REPO_ROOT/tools/r04/mesh_stats.py > $O/ht.log 2>&1; echo "host_trap rc=$?"; tail -3 $O/ht.log
find /tmp/pcs_ht -type f | head
fi
for f in $(find /tmp/pcs_st /tmp/pcs_ht -name "*pc_sampling*" 2>/dev/null); do ls -la $f; head -3 $f; cp $f $O/ 2>/dev/null; done
ls -la $O; du -sh $O
