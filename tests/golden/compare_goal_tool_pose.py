"""``GoalToolPose.from_poses`` (dictionary of per-frame poses -> the [batch, horizon, frames, goal set, 3 | 4] goal tensors the pose
cost reads) against the reference's (types/tool_pose.py) on the CPU: frame order, goal-set layout, padding.

    python tests/golden/compare_goal_tool_pose.py        (needs /root/reference)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401
import torch  # noqa: E402
from curobo._src.types.pose import Pose as RefPose  # noqa: E402
from curobo._src.types.tool_pose import GoalToolPose as Ref  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.types import GoalToolPose as Ours  # noqa: E402
from curobo_amd.types import Pose as OurPose  # noqa: E402

rng = np.random.default_rng(8)
ok = True
for label, b, g, frames, order in (("one frame, batch 3", 3, 1, ["tool0"], ["tool0"]), ("two frames, given in the other order", 2, 1, ["tool1", "tool0"], ["tool0", "tool1"]),
                                   ("goal set of 4, batch 2", 2, 4, ["hand"], ["hand"]), ("goal set of 3, two frames", 1, 3, ["a", "b"], ["b", "a"])):
    data = {}
    for f in frames:
        p = rng.uniform(-1, 1, (b * g, 3)).astype(np.float32)
        q = rng.normal(size=(b * g, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        data[f] = (p, q)  # [batch * goal set, 3 | 4], goal-set members of a problem adjacent
    try:
        r = Ref.from_poses({f: RefPose(position=torch.as_tensor(p), quaternion=torch.as_tensor(q)) for f, (p, q) in data.items()}, ordered_tool_frames=order, num_goalset=g)
        o = Ours.from_poses({f: OurPose(position=torch.as_tensor(p), quaternion=torch.as_tensor(q)) for f, (p, q) in data.items()}, ordered_tool_frames=order, num_goalset=g)
        same = tuple(r.position.shape) == tuple(o.position.shape) and np.array_equal(r.position.numpy(), o.position.numpy()) and np.array_equal(r.quaternion.numpy(), o.quaternion.numpy()) \
            and list(r.tool_frames) == list(o.tool_frames)
        print(f"{label}: {'ok' if same else 'DIFFERENT'}  shape {tuple(r.position.shape)} / {tuple(o.position.shape)}  frames {list(r.tool_frames)} / {list(o.tool_frames)}")
        ok &= same
    except Exception as e:  # noqa: BLE001
        ok = False
        print(f"{label}: ERROR {type(e).__name__}: {str(e)[:300]}")
# the members ToolPose and GoalToolPose share (get_link_pose, to_dict, indexing, reorder_links, as_goal, copies): types/tool_pose.py:54-357
from curobo._src.types.tool_pose import ToolPose as RefTool  # noqa: E402

from curobo_amd.kinematics import ToolPose as OurTool  # noqa: E402

frames = ["a", "b", "c"]
for label, R_, O_, shape in (("ToolPose members", RefTool, OurTool, (4, 3, 3)), ("GoalToolPose members", Ref, Ours, (4, 3, 3, 2))):
    p, q = torch.as_tensor(rng.normal(size=(*shape, 3)).astype(np.float32)), torch.as_tensor(rng.normal(size=(*shape, 4)).astype(np.float32))
    r, o = R_(list(frames), p.clone(), q.clone()), O_(list(frames), p.clone(), q.clone())

    def eq(a, b):
        return tuple(a.position.shape) == tuple(b.position.shape) and torch.equal(a.position, b.position) and torch.equal(a.quaternion, b.quaternion) \
            and list(getattr(a, "tool_frames", [])) == list(getattr(b, "tool_frames", []))

    checks = [eq(r.get_link_pose("b"), o.get_link_pose("b")), eq(r["c"], o["c"]), eq(r[2], o[2]), eq(r[torch.tensor([0, 3])], o[torch.tensor([0, 3])]),
              eq(r.reorder_links(["c", "a"]), o.reorder_links(["c", "a"])), r.reorder_links(frames) is r and o.reorder_links(frames) is o,
              eq(r.clone(), o.clone()), eq(r.detach(), o.detach()), len(r) == len(o), r.ndim == o.ndim, tuple(r.shape) == tuple(o.shape),
              (r.batch_size, r.horizon, r.num_links) == (o.batch_size, o.horizon, o.num_links),
              all(eq(a, b) for a, b in zip(r.to_dict().values(), o.to_dict().values())) and list(r.to_dict()) == list(o.to_dict())]
    if R_ is RefTool:
        checks += [eq(r.as_goal(), o.as_goal()), eq(r.as_goal(["b", "a"]), o.as_goal(["b", "a"])), eq(r.contiguous(), o.contiguous())]
    r2, o2 = R_(list(frames), p * 0, q * 0), O_(list(frames), p * 0, q * 0)
    r2.copy_(r), o2.copy_(o)
    checks.append(eq(r2, o2))
    for bad in (lambda x: x.get_link_pose("zz"), lambda x: x.reorder_links(["a", "zz"])):
        got = []
        for x in (r, o):
            try:
                bad(x)
                got.append(False)
            except Exception:  # noqa: BLE001
                got.append(True)
        checks.append(all(got))
    good = all(checks)
    ok &= good
    print(f"{label}: {'ok' if good else 'DIFFERENT ' + str(checks)}")
sys.exit(0 if ok else 1)
