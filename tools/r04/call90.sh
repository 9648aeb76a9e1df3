#!/bin/bash
# the same test with the guard switched off: the two-level fork inside one capture (expected: the process dies in hipStreamEndCapture)
cd "$GRAFT_REPO_ROOT"
timeout 200 python -X faulthandler - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import sys, torch
sys.path.insert(0, "tests")
import curobo_amd.rollout.trajopt_rollout as T
T.inside_forked_stream = lambda: False
import test_gpu_trajopt as G
G.test_seed_shards_over_torque_limited_rollouts_capture_and_match_one_batch(torch.device("cuda:0"))
print("completed WITHOUT the guard")
PY
echo "exit status: $?"
