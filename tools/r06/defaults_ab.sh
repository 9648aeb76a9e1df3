# A/B of the two defaults the round-5 review names, on the GPU box: seed-knot placement (planner benchmark, both worlds) and the MPC task values
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c
for p in even reference; do
  CUROBO_SEED_KNOT_PLACEMENT=$p timeout 600 python tools/r05/planner_benchmark.py ${1:-100} gpurun_out/r06c/planner_benchmark_seed_knots_$p.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln.rstrip()[:300]); continue
    print('$p', d['world'], 'success', d['success_percent'], 'median ms', round(d['plan_ms']['median'],2), 'p98', round(d['plan_ms']['p98'],1), 'motion s', round(d['motion_s_mean'] or 0,3), d['failures'])
"
done
timeout 600 python tools/r06/mpc_task_compare.py gpurun_out/r06c/mpc_task_compare.json 2>&1 | grep -v amdgpu.ids | tail -6
