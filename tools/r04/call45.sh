#!/bin/bash
# round 4 (ABI 5): the reference's own tests and wrappers over libcurobo_hip.so on the MI355X
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/reference_on_hip.py run > gpurun_out/ref_on_hip_run.log 2>&1; tail -15 gpurun_out/ref_on_hip_run.log
ls gpurun_out/ref_on_hip | head
