"""Generate the robot-model fixtures shipped with the package.

Runs only where the reference checkout is available (this build container): parses the
reference's robot YAML + URDF (read-only data under /root/reference/curobo/content) with
curobo_amd.robot.loader and writes the flat kernel tensors to curobo_amd/content/robot/*.npz.
The GPU box has no /root/reference, so tests / bench / smoke load these .npz files.

Also cross-checks the self-collision pair list against the reference's own (importable)
``SelfCollisionKinematicsCfg.create_from_link_pairs``.
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CUROBO_REFERENCE", "/root/reference")
CONTENT = os.path.join(REF, "curobo", "content")

from curobo_amd.robot import load_robot_model  # noqa: E402


def reference_pairs(model, cfg_yaml):
    """Pair list from the reference implementation itself (pure torch, imports on CPU)."""
    import torch
    import yaml

    sys.path.insert(0, REF)
    from curobo._src.robot.types.self_collision_params import SelfCollisionKinematicsCfg
    from curobo._src.types.device_cfg import DeviceCfg

    data = yaml.safe_load(open(cfg_yaml))
    data = data.get("robot_cfg", data)["kinematics"]
    name_to_idx = {n: i for i, n in enumerate(model.link_names)}
    cfg = SelfCollisionKinematicsCfg.create_from_link_pairs(
        collision_link_names=list(data["collision_link_names"]),
        link_name_to_sphere_index=name_to_idx,
        self_collision_link_pair_ignores=data.get("self_collision_ignore") or {},
        self_collision_link_padding=dict(data.get("self_collision_buffer") or {}),
        all_link_spheres=torch.as_tensor(model.link_spheres[0]),
        link_index_to_sphere_index=torch.as_tensor(model.link_sphere_idx_map),
        device_cfg=DeviceCfg(device="cpu"),
    )
    return cfg.collision_pairs.numpy(), cfg.sphere_padding.numpy()


def main():
    out_dir = os.path.join(ROOT, "curobo_amd", "content", "robot")
    os.makedirs(out_dir, exist_ok=True)
    for name in ("franka", "ur10e", "dual_ur10e", "unitree_g1"):
        yml = os.path.join(CONTENT, "configs", "robot", f"{name}.yml")
        model = load_robot_model(yml, os.path.join(CONTENT, "assets"))
        if name != "unitree_g1":  # the O(S^2) python loop of the reference takes minutes on G1
            pairs, padding = reference_pairs(model, yml)
            assert np.array_equal(pairs, model.collision_pairs), f"{name}: pair list differs from reference"
            assert np.allclose(padding, model.sphere_padding), f"{name}: padding differs from reference"
            print(f"{name}: self-collision pair list identical to the reference ({len(pairs)} pairs)")
        path = os.path.join(out_dir, f"{name}.npz")
        model.save_npz(path)
        print(f"{name}: D={model.num_dof} L={model.num_links} S={model.num_spheres} "
              f"T={len(model.tool_frames)} P={model.collision_pairs.shape[0]} -> {path} "
              f"({os.path.getsize(path) / 1024:.0f} KiB)")


def kinematics_only():
    """The robot files as the reference's IK benchmark loads them for its `IK` rows (benchmark/ik_benchmark.py:60-65: collision_link_names
    = None, lock_joints = None): no spheres, and only the joints on the chains to the tool frames -> <name>_kinematics_only.npz"""
    import tempfile

    import yaml

    out_dir = os.path.join(ROOT, "curobo_amd", "content", "robot")
    for name in ("franka", "dual_ur10e", "unitree_g1"):
        yml = os.path.join(CONTENT, "configs", "robot", f"{name}.yml")
        data = yaml.safe_load(open(yml))
        k = data.get("robot_cfg", data)["kinematics"]
        k["collision_link_names"], k["lock_joints"] = None, None
        with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as f:
            yaml.safe_dump(data, f)
        model = load_robot_model(f.name, os.path.join(CONTENT, "assets"))
        os.unlink(f.name)
        if model.num_spheres == 0:
            # one DISABLED sphere (negative radius, the convention of the attached-object placeholders) on the base link: the sphere
            # buffers of every launch keep a non-empty shape, and no kernel counts a disabled sphere
            import dataclasses

            E = model.link_spheres.shape[0]
            model = dataclasses.replace(
                model, link_spheres=np.tile(np.array([[[0.0, 0.0, 0.0, -10.0]]], np.float32), (E, 1, 1)),
                link_sphere_idx_map=np.zeros(1, model.link_sphere_idx_map.dtype), sphere_padding=np.zeros(1, np.float32),
                collision_pairs=np.zeros((0, 2), model.collision_pairs.dtype))
        path = os.path.join(out_dir, f"{name}_kinematics_only.npz")
        model.save_npz(path)
        print(f"{name}_kinematics_only: D={model.num_dof} L={model.num_links} S={model.num_spheres} T={len(model.tool_frames)} -> {path}")


if __name__ == "__main__":
    if "--kinematics-only" in sys.argv:
        kinematics_only()
    else:
        main()
        kinematics_only()
