#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for p in 0 -1 1; do echo "== priority $p"; CUROBO_SIDE_STREAM_PRIORITY=$p timeout 200 python tools/c4_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tail -4; done
