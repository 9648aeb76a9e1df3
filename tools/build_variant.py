"""Build libcurobo_hip.so variants with extra compiler flags for ONE source (tuning experiments).

    python tools/build_variant.py <name> <source.hip> <flag> [<flag> ...]

compiles <source.hip> with the package's flags + the extra ones, links it with the package's other (already built)
objects into curobo_amd/lib/variants/libcurobo_hip_<name>.so.  To measure one on the GPU box, copy it over
curobo_amd/lib/libcurobo_hip.so inside the gpurun command (the box works on a scratch copy of the repository).
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from curobo_amd import build as B  # noqa: E402


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()  # the other objects
    out_dir = os.path.join(B.LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"{src.rsplit('.', 1)[0]}_{name}.o")
    cmd = [B.hipcc_path(), *B._flags(), *B.NO_SLP, *extra, "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    objs = [obj if s == src else os.path.join(B.OBJ_DIR, s.rsplit(".", 1)[0] + ".o") for s in B.SOURCES]
    lib = os.path.join(out_dir, f"libcurobo_hip_{name}.so")
    subprocess.check_call([B.hipcc_path(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
    print(lib)


if __name__ == "__main__":
    main()
