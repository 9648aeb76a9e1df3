"""Static instruction counts of sphere_mesh_walk_kernel<3> by source region (device assembly with line tables; no GPU needed).

    python tools/r04/walk_isa_breakdown.py            -> table on stdout
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from curobo_amd import build as B  # noqa: E402

asm = os.path.join(tempfile.mkdtemp(), "mesh.s")
subprocess.run([B.hipcc_path(), *B._flags(), *B.NO_SLP, "-gline-tables-only", "--cuda-device-only", "-S", "-x", "hip",
                os.path.join(B.CSRC, "mesh_bvh.hip"), "-o", asm], check=True, capture_output=True)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN10curobo_hip23sphere_mesh_walk_kernelILi3EEEvNS_13MeshQueueArgsE:"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
src = open(os.path.join(B.CSRC, "mesh_device.hpp")).read().split("\n")
mark = lambda text: next(i + 1 for i, l in enumerate(src) if text in l)  # noqa: E731
bounds = [("walker set-up, begin_query", mark("__device__ __forceinline__ void mesh_contribution_group")),
          ("leaf move", mark("if (d0 == depth_leaves) {  // a leaf")), ("interior move", mark("} else {  // an interior node")),
          ("sibling / up", mark("if (!descended) {")), ("settle a query", mark("the walk of a query is over")),
          ("terms, next sample", mark("its terms, and the next sample")), ("(end)", mark("__device__ __forceinline__ f3 mesh_to_world_vector"))]
cnt = collections.defaultdict(collections.Counter)
cur = ("?", 0)
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    op = l.strip().split(" ")[0] if l.strip() else ""
    if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
        continue
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    f, ln = cur
    if f == "mesh_device.hpp" and ln >= bounds[0][1] and ln < bounds[-1][1]:
        region = [n for n, b in bounds if ln >= b][-1]
    elif ln == 0:
        region = "no line (control flow the compiler adds: exec masks, loop structure)"
    elif f == "mesh_device.hpp":
        region = "helpers of mesh_device.hpp (closest point, box distance, feature side, inside test)"
    elif f == "mesh_bvh.hip":
        region = "the kernel body (queue entry, sphere prologue, slot loop, outputs)"
    else:
        region = f"inlined from {f}"
    cnt[region][kind] += 1
tot = collections.Counter()
for region, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
    tot.update(c)
    print(f"{sum(c.values()):5d}  valu {c['valu']:4d} salu {c['salu']:4d} vmem {c['vmem']:3d} lds {c['lds']:3d}   {region}")
print(f"{sum(tot.values()):5d}  valu {tot['valu']:4d} salu {tot['salu']:4d} vmem {tot['vmem']:3d} lds {tot['lds']:3d}   total (static)")
