"""bitwise comparison of the packed scene-collision kernel with the in-lane obstacle loop (CUROBO_HIP_SCENE_UNPACKED=1):
   python tools/scene_packed_vs_unpacked.py run out.npz   (in two processes), then  ... cmp a.npz b.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = all(np.array_equal(a[k], b[k]) for k in a.files)
    print("bit-identical:", ok, {k: float(np.abs(a[k] - b[k]).max()) for k in a.files})
    sys.exit(0 if ok else 1)
import torch
from conftest import load_model, sample_q
from oracle import load_oracle
from curobo_amd.backends import collision as Cn
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world, c3_voxel_world
dev = torch.device("cuda:0"); orc = load_oracle(); out = {}
for name, robot, arrays in (("cub", "franka", cuboid_scene_arrays(c2_world())), ("vox", "ur10e", c3_voxel_world(64, 0.04)),
                            ("mix", "franka", {**cuboid_scene_arrays(c2_world()), **c3_voxel_world(64, 0.04)})):
    model = load_model(robot); b, h = 48, 17
    q0, q1 = sample_q(model, b, seed=3)[:, None], sample_q(model, b, seed=4)[:, None]
    tt = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    sph = orc.kinematics_forward((q0 * (1 - tt) + q1 * tt).reshape(b * h, -1) * 0.7, model.as_dict(), horizon=h)["robot_spheres"].reshape(b, h, -1, 4)
    S = sph.shape[2]; scene = SceneData.from_arrays(arrays, dev)
    for sweep in (0, 3):
        dist, grad = torch.zeros(b, h, S, device=dev), torch.zeros(b, h, S, 4, device=dev)
        Cn.sphere_obstacle_collision(dist, grad, torch.as_tensor(sph, device=dev), scene.struct, torch.tensor([7.0], device=dev),
                                     torch.tensor([0.02], device=dev), None, b, h, S, False, sweep, sweep > 0, torch.tensor([0.05], device=dev))
        torch.cuda.synchronize()
        out[f"{name}{sweep}_d"], out[f"{name}{sweep}_g"] = dist.cpu().numpy(), grad.cpu().numpy()
        print(name, sweep, "hit fraction", float((dist > 0).float().mean()))
np.savez(sys.argv[2], **out)
