#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call20; mkdir -p $O
for bd in 4,3 2,3 1,3 2,2 1,2 8,3; do echo "== coarse block,dilate $bd"; CUROBO_VOXEL_COARSE=$bd python bench.py --only c3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3_ur10e_voxel']
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('us','us_per_launch')}) for k,v in d.items() if k in ('fused_launch_us','kernel_sequence_us','kernels','us_per_rollout_set','value','scene_kernel_us')})
print(json.dumps(d)[:600])
"; done | tee $O/coarse.txt
timeout 300 python -m pytest tests/test_public_api.py -q -m gpu -k "lbfgs or LBFGS or optim" 2>&1 | tail -3
