"""Which spheres make the longest walks of the bench's mesh world (needs the -DCUROBO_MESH_STATS build): the heaviest queue
entries with their sphere, neighbours and live-slot mask -> gpurun_out/mesh_heavy.json"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("SUBDIV", "4")
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("mesh_ab", os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh_ab.py"))
sys.argv = [sys.argv[0]]
ab = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ab)  # builds the world, runs and times the launches
from curobo_amd.backends import mesh as M  # noqa: E402

lib, ro, B, S, H = ab.lib, ab.ro, ab.B, ab.S, 33
out = {"world": [dict(pose=[float(v) for v in o["pose"]], dims=[float(v) for v in o["dims"]]) for o in ab.world[0]]}
for sweep in (0, 3):
    lib.curobo_hip_mesh_lane_stats(None, 1)
    ab.scene_pass(sweep)
    torch.cuda.synchronize()
    lanes = np.zeros(1 << 18, np.uint32)
    lib.curobo_hip_mesh_lane_stats(lanes.ctypes.data_as(C.c_void_p), 1)
    moves, trans = lanes[:1 << 17].astype(np.int64), lanes[1 << 17:].astype(np.int64)
    ws = next(iter(M._WORKSPACES.values())).cpu().numpy()
    cnt = ws[:16].view(np.uint32)
    n_front, n_back = int(cnt[0]), int(cnt[2])
    queue = ws[16:].view(np.uint32).reshape(-1, 2)
    total = B * H * S
    sph = ro.robot_spheres.detach().cpu().numpy().reshape(-1, 4)
    top = np.argsort(moves + trans)[::-1][:12]
    items = []
    for q in top:
        q = int(q)
        e = queue[q] if q < n_front else queue[total - 1 - (q - n_front)]
        sidx = int(e[0])
        b, rem = divmod(sidx, H * S)
        h, s = divmod(rem, S)
        items.append(dict(queue_index=q, heavy_class=bool(q < n_front), moves=int(moves[q]), transitions=int(trans[q]), batch=b, point=h, sphere=s,
                          mask=int(e[1]), xyzr=[float(v) for v in sph[sidx]],
                          prev=[float(v) for v in sph[sidx - S]] if h > 0 else None, next=[float(v) for v in sph[sidx + S]] if h < H - 1 else None))
    out[f"sweep{sweep}"] = dict(n_front=n_front, n_back=n_back, heaviest=items)
    print(f"sweep {sweep}: heavy class {n_front}, others {n_back}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/mesh_heavy.json", "w"), indent=1)
