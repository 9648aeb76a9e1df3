#!/bin/bash
# tools/r06/build_mesh_variant.sh <name> <flag> ...: mesh_bvh.hip compiled with extra flags, linked with the package's other
# objects into curobo_amd/lib/variants/libcurobo_hip_<name>.so (copy it over curobo_amd/lib/libcurobo_hip.so on the GPU box)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p curobo_amd/lib/variants
python - "$name" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from curobo_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
B.build()
obj = os.path.join(B.LIB_DIR, "variants", f"mesh_bvh_{name}.o")
subprocess.check_call([B.hipcc_path(), *B._flags(), *B.NO_SLP, *extra, "-x", "hip", "-c", os.path.join(B.CSRC, "mesh_bvh.hip"), "-o", obj,
                       "-Rpass-analysis=kernel-resource-usage"], stderr=open(obj + ".log", "w"))
objs = [obj if stem == "mesh_bvh" else os.path.join(B.OBJ_DIR, stem + ".o") for _, stem, _ in B.compile_units()]
lib = os.path.join(B.LIB_DIR, "variants", f"libcurobo_hip_{name}.so")
subprocess.check_call([B.hipcc_path(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
log = open(obj + ".log").read().split("Function Name")
for blk in log:
    if "cells_kernelILi3" in blk:
        print(blk.split("[")[0].strip()[:90])
        import re
        print(name, re.findall(r"(VGPRs: \d+|VGPRs Spill: \d+|Occupancy \[waves/SIMD\]: \d+)", blk))
os.remove(obj)
PY
