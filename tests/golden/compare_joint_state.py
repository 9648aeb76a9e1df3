"""``curobo_amd.types.JointState`` members against the reference's ``JointState`` (state/state_joint.py, state_joint_ops.py,
state_joint_trajectory_ops.py) on random states, member by member.
    python tests/golden/compare_joint_state.py        (needs /root/reference)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401
import torch  # noqa: E402
from curobo._src.state.state_joint import JointState as Ref  # noqa: E402
from curobo._src.types.device_cfg import DeviceCfg as RefDeviceCfg  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.types import DeviceCfg, JointState as Ours  # noqa: E402

rng = np.random.default_rng(11)
t = torch.as_tensor
cpu_ref, cpu_ours = RefDeviceCfg(device=torch.device("cpu")), DeviceCfg(device=torch.device("cpu"))
names = [f"j{i}" for i in range(6)]
ok = True


def rnd(*shape):
    return t(rng.normal(size=shape).astype(np.float32))


def pair(*shape, dt=None, with_names=True):
    p, v, a, j = rnd(*shape), rnd(*shape), rnd(*shape), rnd(*shape)
    nm = list(names[:shape[-1]]) if with_names else None
    d = None if dt is None else t(np.asarray(dt, dtype=np.float32))
    mk = lambda cls: cls(position=p.clone(), velocity=v.clone(), acceleration=a.clone(), jerk=j.clone(), joint_names=None if nm is None else list(nm),  # noqa: E731
                         dt=None if d is None else d.clone())
    return mk(Ref), mk(Ours)


def same(r, o, tol=0.0):
    """the four tensors, the names and dt of two states"""
    for f in ("position", "velocity", "acceleration", "jerk", "dt"):
        a, b = getattr(r, f), getattr(o, f)
        if (a is None) != (b is None):
            return f"{f}: None on one side"
        if a is not None:
            if tuple(a.shape) != tuple(b.shape):
                return f"{f}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
            if a.numel() and float((a.double() - b.double()).abs().max()) > tol:
                return f"{f}: values"
    if (r.joint_names or None) != (o.joint_names or None):
        return f"names {r.joint_names} vs {o.joint_names}"
    return ""


def check(what, r, o, tol=0.0):
    global ok
    why = same(r, o, tol) if not isinstance(r, torch.Tensor) else ("" if r.shape == o.shape and float((r - o).abs().max()) <= tol else "values")
    ok &= why == ""
    print(f"{what}: {'ok' if why == '' else 'DIFFERENT -- ' + why}")


# constructors
pos = rnd(5, 6)
check("from_position", Ref.from_position(pos, list(names)), Ours.from_position(pos, list(names)))
check("from_numpy", Ref.from_numpy(list(names), pos.numpy(), velocity=pos.numpy() * 2, device_cfg=cpu_ref),
      Ours.from_numpy(list(names), pos.numpy(), velocity=pos.numpy() * 2, device_cfg=cpu_ours))
st = rnd(4, 9, 24)
check("from_state_tensor", Ref.from_state_tensor(st, list(names), dof=6), Ours.from_state_tensor(st, list(names), dof=6))
check("from_list", Ref.from_list([[0.1, 0.2]], [[0.3, 0.4]], [[0.5, 0.6]], cpu_ref), Ours.from_list([[0.1, 0.2]], [[0.3, 0.4]], [[0.5, 0.6]], cpu_ours))
check("zeros", Ref.zeros((3, 6), cpu_ref, list(names)), Ours.zeros((3, 6), cpu_ours, list(names)))
# shape members
r, o = pair(4, 9, 6, dt=[0.1, 0.2, 0.3, 0.4])
check("get_state_tensor", r.get_state_tensor(), o.get_state_tensor())
check("clone", r.clone(), o.clone())
check("detach", r.detach(), o.detach())
check("unsqueeze", r.unsqueeze(1), o.unsqueeze(1))
check("squeeze", r.unsqueeze(0).squeeze(0), o.unsqueeze(0).squeeze(0))
check("getitem int", r[2], o[2])
check("getitem tensor", r[t([0, 3])], o[t([0, 3])])
check("getitem slice", r[1:3], o[1:3])
check("getitem list", r[[1, 2]], o[[1, 2]])
ok &= (len(r) == len(o)) and tuple(r.shape) == tuple(o.shape) and r.ndim == o.ndim
r2, o2 = pair(4, 9, 6)
check("stack", r2.stack(r2), o2.stack(o2))
r3, o3 = pair(4, 9, 6)
check("cat dim 0", r2.cat(r3, 0), o2.cat(o3, 0))
rb, ob = pair(5, 6)
check("repeat", rb.repeat([3, 1]), ob.repeat([3, 1]))
check("repeat_seeds", rb.repeat_seeds(4), ob.repeat_seeds(4))
check("view", rb.repeat_seeds(4).view(5, 4, 6), ob.repeat_seeds(4).view(5, 4, 6))
ker = rnd(7, 5)
check("apply_kernel", rb.apply_kernel(ker), ob.apply_kernel(ker), tol=1e-6)
# time scaling and finite differences
check("scale", rb.scale(0.5), ob.scale(0.5), tol=1e-7)
r, o = pair(4, 9, 6, dt=[0.1, 0.2, 0.3, 0.4])
new_dt = t(np.asarray([0.05, 0.4, 0.3, 0.2], dtype=np.float32))
check("scale_by_dt 3-d", r.scale_by_dt(r.dt, new_dt), o.scale_by_dt(o.dt, new_dt), tol=1e-6)
check("scale_time", r.scale_time(new_dt), o.scale_time(new_dt), tol=1e-6)
r1, o1 = pair(9, 6, dt=[0.1])
check("scale_by_dt 2-d", r1.scale_by_dt(r1.dt, t([0.25])), o1.scale_by_dt(o1.dt, t([0.25])), tol=1e-6)
r, o = pair(4, 9, 6, dt=[0.05])
check("calculate_fd_from_position", r.calculate_fd_from_position(), o.calculate_fd_from_position(), tol=1e-3)
r, o = pair(4, 9, 6, dt=[[0.1], [0.2], [0.3], [0.4]])
check("calculate_fd_from_position, dt per trajectory", r.calculate_fd_from_position(), o.calculate_fd_from_position(), tol=1e-3)
# joint bookkeeping
order = ["j3", "j0", "j5", "j1", "j2", "j4"]
r, o = pair(4, 6)
check("reorder", r.reorder(order), o.reorder(order))
check("reorder subset", r.reorder(order[:3]), o.reorder(order[:3]))
r.reindex(order), o.reindex(order)
check("reindex", r, o)
r, o = pair(4, 6)
check("index_dof", r.index_dof(t([4, 1])), o.index_dof(t([4, 1])))
lock_p = rnd(2)
lr = Ref.from_position(lock_p.clone(), ["gripper_l", "gripper_r"])
lo = Ours.from_position(lock_p.clone(), ["gripper_l", "gripper_r"])
full = ["gripper_r"] + order + ["gripper_l"]
check("append_joints 2-d", r.clone().append_joints(lr), o.clone().append_joints(lo))
check("get_augmented_joint_state", r.get_augmented_joint_state(full, lr), o.get_augmented_joint_state(full, lo))
check("get_augmented_joint_state (no lock joints)", r.get_augmented_joint_state(order), o.get_augmented_joint_state(order))
r3, o3 = pair(3, 5, 6)
ra3 = r3.clone().append_joints(lr)
ra3.dt = None  # (the reference's >= 3-d branch builds on JointState.zeros and so hands back a dt of ones; ours keeps the state's dt)
check("append_joints 3-d", ra3, o3.clone().append_joints(lo))
r1d, o1d = pair(6)
check("append_joints 1-d", r1d.clone().append_joints(lr), o1d.clone().append_joints(lo))
for what, fn in (("append_joints: names twice", lambda s, l: s.get_augmented_joint_state(full, type(s).from_position(lock_p, ["j0", "x"]))),
                 ("reorder: unknown joint", lambda s, l: s.reorder(["j0", "nope"]))):
    got = []
    for s, l in ((r, lr), (o, lo)):
        try:
            fn(s, l)
            got.append("no error")
        except Exception as e:  # noqa: BLE001
            got.append("error")
    good = got[0] == got[1] == "error"
    ok &= good
    print(f"{what}: {'ok' if good else 'DIFFERENT -- ' + str(got)}")
# seeds, trajectories, copies
r, o = pair(3, 5, 7, 6, dt=rng.uniform(0.01, 0.1, (3, 5)))
idx = t(np.asarray([[4, 0], [1, 1], [2, 3]]))
check("gather_by_seed_index", r.gather_by_seed_index(idx), o.gather_by_seed_index(idx))
r, o = pair(4, 9, 6)
check("get_trajectory_at_horizon_index", r.get_trajectory_at_horizon_index(-1), o.get_trajectory_at_horizon_index(-1))
check("trim_trajectory", r.trim_trajectory(2, 7), o.trim_trajectory(2, 7))
check("trim_trajectory to the end", r.trim_trajectory(3), o.trim_trajectory(3))
src_r, src_o = pair(4, 9, 6)
check("copy_", r.clone().copy_(src_r), o.clone().copy_(src_o))
small_r, small_o = pair(2, 6)
check("copy_ of another shape", r.clone().copy_(small_r), o.clone().copy_(small_o))
check("copy_data", r.clone().copy_data(src_r), o.clone().copy_data(src_o))
check("copy_only_index", r.clone().copy_only_index(src_r, t([1, 3])), o.clone().copy_only_index(src_o, t([1, 3])))
cr, co = r.clone(), o.clone()
cr.copy_at_index(src_r[t([0, 1])], t([2, 3])), co.copy_at_index(src_o[t([0, 1])], t([2, 3]))
check("copy_at_index", cr, co)
cr, co = r.clone(), o.clone()
cr[t([0, 2])] = src_r[t([1, 3])]
co[t([0, 2])] = src_o[t([1, 3])]
check("setitem", cr, co)
r, o = pair(3, 5, 7, 6)
src_r, src_o = pair(3, 5, 7, 6)
bi, si = t([0, 2, 2]), t([1, 0, 4])
check("copy_at_batch_seed_indices", r.clone().copy_at_batch_seed_indices(src_r, bi, si), o.clone().copy_at_batch_seed_indices(src_o, bi, si))
rr = Ref.from_position(rnd(2, 6), list(names))
rr.copy_reference(src_r)
oo = Ours.from_position(rnd(2, 6), list(names))
oo.copy_reference(src_o)
check("copy_reference", rr, oo)
check("to", r.to(cpu_ref), o.to(cpu_ours))
print("all ok" if ok else "DIFFERENCES")
sys.exit(0 if ok else 1)
