"""``curobo_amd.types.ToolPoseCriteria`` against the reference's (cost/tool_pose_criteria.py) run on the CPU: every factory
with its defaults and with arguments -> the same axis factors, tolerances and projection flag.

    python tests/golden/compare_tool_pose_criteria.py        (needs /root/reference)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401
from curobo._src.cost.tool_pose_criteria import ToolPoseCriteria as Ref  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.types import ToolPoseCriteria as Ours  # noqa: E402

CASES = [("track_position", (), {}), ("track_position", ([1.0, 0.0, 0.5],), {}), ("track_orientation", (), {}),
         ("track_orientation", ([0.1, 0.2, 0.3],), {"non_terminal_scale": 0.5}), ("track_position_and_orientation", (), {}),
         ("track_position_and_orientation", ([1, 0, 1], [0, 1, 0]), {"non_terminal_scale": 0.3}), ("linear_motion", (), {}),
         ("linear_motion", ("x",), {"non_terminal_scale": 2.0}), ("linear_motion", ("y",), {"project_distance_to_goal": False}), ("disabled", (), {})]
FIELDS = ("terminal_pose_axes_weight_factor", "non_terminal_pose_axes_weight_factor", "terminal_pose_convergence_tolerance",
          "non_terminal_pose_convergence_tolerance", "project_distance_to_goal")
ok = True
for name, a, kw in CASES:
    r, o = getattr(Ref, name)(*a, **kw), getattr(Ours, name)(*a, **kw)
    for f in FIELDS:
        rv = np.asarray(getattr(r, f).detach().cpu().numpy() if hasattr(getattr(r, f), "detach") else getattr(r, f), np.float64).reshape(-1)
        ov = np.asarray(getattr(o, f), np.float64).reshape(-1)
        if rv.shape != ov.shape or not np.allclose(rv, ov, rtol=0, atol=1e-6):
            ok = False
            print(f"DIFFERENT {name}{a}{kw} {f}: {rv} != {ov}")
print(f"{len(CASES)} factory calls: {'ok' if ok else 'DIFFERENT'}")
sys.exit(0 if ok else 1)
