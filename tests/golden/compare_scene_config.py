"""The cuboid store a scene description becomes (dims, inverse poses, enable flags, counts: the obstacle input of the
collision kernels) -- ``curobo_amd.scene.config.scene_arrays_from_config`` against THE REFERENCE'S OWN ``SceneCfg.create`` +
``CuboidData.from_scene_cfg`` / ``from_batch_scene_cfg`` (geom/types.py, geom/data/data_cuboid.py:67-260) run on the CPU, on the
scene files the reference ships and on random rotated cuboids in two environments.

    python tests/golden/compare_scene_config.py        (needs /root/reference)

The pad column of ``dims`` is not compared (the reference leaves its cache default 0.01 there; no kernel reads it)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_robot_loader as R  # noqa: E402,F401  (CPU DeviceCfg, the Warp stand-in, /root/reference on the path)
import yaml  # noqa: E402
from curobo._src.geom.data.data_cuboid import CuboidData  # noqa: E402
from curobo._src.geom.types import SceneCfg  # noqa: E402
from curobo._src.types.device_cfg import DeviceCfg  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from curobo_amd.scene.config import scene_arrays_from_config  # noqa: E402


def compare(label, cfgs):
    cfgs = [{"cuboid": c.get("cuboid") or {}} for c in cfgs]
    scenes = [SceneCfg.create(c) for c in cfgs]
    ref = CuboidData.from_batch_scene_cfg(scenes, DeviceCfg(device="cpu")) if len(scenes) > 1 else CuboidData.from_scene_cfg(scenes[0], DeviceCfg(device="cpu"))
    ours = scene_arrays_from_config(cfgs if len(cfgs) > 1 else cfgs[0])
    bad = []
    if ref.count.tolist() != ours["cuboid_count"].tolist():
        bad.append(f"count {ref.count.tolist()} != {ours['cuboid_count'].tolist()}")
    for e, n in enumerate(ref.count.tolist()):
        for rk, ok, cols in (("dims", "cuboid_dims", 3), ("inv_pose", "cuboid_inv_pose", 7), ("enable", "cuboid_enable", None)):
            a, b = getattr(ref, rk).numpy()[e, :n], ours[ok][e, :n]
            if cols is not None:
                a, b = a[..., :cols], b[..., :cols]
            if not np.allclose(a.astype(np.float64), b.astype(np.float64), rtol=0, atol=1e-6):
                bad.append(f"env {e} {rk}: max |diff| {np.abs(a.astype(np.float64) - b).max():.3e}")
    print(f"{label}: {'ok' if not bad else 'DIFFERENT: ' + '; '.join(bad)}  (cuboids per environment {ref.count.tolist()})", flush=True)
    return not bad


if __name__ == "__main__":
    d = os.path.join(R.REF, "curobo", "content", "configs", "scene")
    ok = True
    for f in sorted(os.listdir(d)):
        ok &= compare(f, [yaml.safe_load(open(os.path.join(d, f)))])
    rng = np.random.default_rng(4)

    def world(n):
        out = {}
        for i in range(n):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            out[f"box{i}"] = {"dims": [float(v) for v in rng.uniform(0.05, 1.0, 3)], "pose": [float(v) for v in rng.uniform(-1, 1, 3)] + [float(v) for v in q]}
        return {"cuboid": out}

    ok &= compare("random rotated cuboids, two environments (5 and 3)", [world(5), world(3)])

    # ---- voxel grids (fp16 ESDF): VoxelData.from_scene_cfg (geom/data/data_voxel.py:283-470) against voxel_arrays_from_config
    import torch
    from curobo._src.geom.data.data_voxel import VoxelData
    from curobo._src.geom.types import VoxelGrid

    from curobo_amd.scene.config import voxel_arrays_from_config

    def grid(name, dims, vs):
        n = [int(round(d / vs)) for d in dims]
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        f = rng.uniform(-0.2, 0.5, size=int(np.prod(n))).astype(np.float16)
        return name, dict(dims=list(dims), voxel_size=vs, pose=[float(v) for v in rng.uniform(-1, 1, 3)] + [float(v) for v in q], feature_tensor=torch.as_tensor(f))

    cfg = {"voxel": dict([grid("a", (0.5, 0.4, 0.3), 0.02), grid("b", (0.3, 0.3, 0.6), 0.05)])}
    grids = [VoxelGrid(name=k, pose=v["pose"], dims=v["dims"], voxel_size=v["voxel_size"], feature_tensor=v["feature_tensor"].clone().view(-1, 1),
                       feature_dtype=torch.float16) for k, v in cfg["voxel"].items()]
    ref = VoxelData.from_scene_cfg(SceneCfg(voxel=grids), DeviceCfg(device="cpu"))
    ours = voxel_arrays_from_config(cfg)
    bad = []
    # (the reference stores dims / voxel_size as computed in fp32, e.g. 15.000001 cells; its kernels truncate it to an integer)
    if not np.array_equal(np.round(ref.params.numpy()[..., :3]), ours["voxel_params"][..., :3]) or not np.allclose(ref.params.numpy()[..., 3], ours["voxel_params"][..., 3]):
        bad.append("params")
    if not np.allclose(ref.inv_pose.numpy()[..., :7], ours["voxel_inv_pose"][..., :7], rtol=0, atol=1e-6):
        bad.append("inv_pose")
    if ref.enable.numpy().tolist() != ours["voxel_enable"].tolist() or ref.count.numpy().tolist() != ours["voxel_count"].tolist():
        bad.append("enable / count")
    for g, v in enumerate(cfg["voxel"].values()):
        n = v["feature_tensor"].numel()
        if not np.array_equal(ref.features.numpy()[0, g, :n, 0], ours["voxel_features"][0, g, :n]):
            bad.append(f"features of grid {g}")
    print(f"two voxel grids (25 x 20 x 15 at 2 cm, 6 x 6 x 12 at 5 cm, rotated): {'ok' if not bad else 'DIFFERENT: ' + ', '.join(bad)}", flush=True)
    ok &= not bad
    # the cell-centre layout the features are stored in, occupied cells, clone (geom/types.py:846-916)
    from curobo_amd.scene.types import VoxelGrid as OurGrid

    bad = []
    for (name, v), rg in zip(cfg["voxel"].items(), grids):
        og = OurGrid(name=name, pose=v["pose"], dims=v["dims"], voxel_size=v["voxel_size"], feature_tensor=v["feature_tensor"].clone())
        for to_world in (False, True):
            a, b = rg.create_xyzr_tensor(transform_to_origin=to_world, device_cfg=DeviceCfg(device="cpu")), og.create_xyzr_tensor(transform_to_origin=to_world)
            if tuple(a.shape) != tuple(b.shape) or float((a - b).abs().max()) > 2e-6:
                bad.append(f"{name}: cell centres (world frame: {to_world})")
        rg.xyzr_tensor, og.xyzr_tensor = rg.create_xyzr_tensor(device_cfg=DeviceCfg(device="cpu")), og.create_xyzr_tensor()
        rg.feature_tensor = rg.feature_tensor.view(-1)
        for thr in (None, 0.1):
            a, b = rg.get_occupied_voxels(thr), og.get_occupied_voxels(thr)
            if tuple(a.shape) != tuple(b.shape) or a.shape[0] == 0 or float((a.float() - b.float()).abs().max()) > 2e-3:
                bad.append(f"{name}: occupied cells (threshold {thr})")
        c = og.clone()
        if c.feature_tensor is og.feature_tensor or not torch.equal(c.feature_tensor, og.feature_tensor) or c.dims != list(og.dims):
            bad.append(f"{name}: clone")
    print(f"voxel grid cell centres / occupied cells / clone: {'ok' if not bad else 'DIFFERENT: ' + ', '.join(bad)}", flush=True)
    ok &= not bad
    sys.exit(0 if ok else 1)
