#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call19; mkdir -p $O
for t in 0 576 640 768 1024; do echo "== threads $t"; CUROBO_HIP_FUSED_THREADS=$t python tools/trajopt_solve_probe.py 2>&1 | grep "ms per solve"; done | tee $O/threads.txt
for t in 0 768 1024; do echo "== profile_fused batch 32 threads $t"; CUROBO_HIP_FUSED_THREADS=$t python tools/profile_fused.py --batch 32 2>&1 | grep "fused launch\|P0\|P1 FK\|P2 costs\|P3"; done | tee -a $O/threads.txt
