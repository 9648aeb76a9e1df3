"""Run the REFERENCE's own tests of its Warp-backed costs and voxel collision on the CPU, through the Warp stand-in.

    python tools/reference_warp_tests_on_cpu.py        # here (needs /root/reference); writes profiles/r02_reference_warp_tests_on_cpu.json

Purpose: `tests/golden/warp_emulator` (the pure-Python stand-in for NVIDIA Warp that produced the `*_warp_golden.npz`
vectors) is itself checked against the reference's expectations: the reference's test files

    tests/_src/cost/test_cost_tool_pose.py     ToolPoseCost: forward values, goal sets, gradients, gradcheck
    tests/_src/cost/test_cost_cspace.py        PositionCSpaceCost / StateCSpaceCost: bounds, regularisation, gradcheck
    tests/_src/geom/sdf/test_voxel_collision.py  sphere / swept-sphere vs fp16 ESDF, kernel-level scenarios

are copied into an untracked scratch directory with the ONE edit that lets them run without a GPU -- the device string
"cuda:0" becomes "cpu" (and `torch.cuda.is_available` answers yes so that their skip marks do not fire) -- and run with
`warp` = the stand-in, third-party modules the reference imports at module level and these tests never touch (trimesh, lxml,
yourdfpy) stubbed.  Everything else -- the cost classes, their autograd functions, the Warp launches, the kernels, the
assertions -- is the reference's.  Nothing of the reference stays in the repository: the scratch directory is removed.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/curobo"
FILES = ["tests/_src/cost/test_cost_tool_pose.py", "tests/_src/cost/test_cost_cspace.py", "tests/_src/geom/sdf/test_voxel_collision.py"]
CONFTEST = '''
import importlib.abc, importlib.machinery, sys
from unittest.mock import MagicMock
import torch


class _Stub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = {"trimesh", "yourdfpy", "lxml"}

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__, m.__name__, m.__spec__, m.__loader__ = [], spec.name, spec, self
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_Stub())
torch.cuda.is_available = lambda: True   # the copies say "cpu" wherever the originals said "cuda:0"
torch.cuda.synchronize = lambda *a, **k: None
'''


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    scratch = tempfile.mkdtemp(prefix="refwarp_", dir=ROOT)
    report = {"what": "the reference's own test files on the CPU through tests/golden/warp_emulator (only edit: 'cuda:0' -> 'cpu')", "files": {}}
    try:
        with open(os.path.join(scratch, "conftest.py"), "w") as f:
            f.write(CONFTEST)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "golden", "warp_emulator"), "/root/reference"]))
        for rel in FILES:
            name = os.path.basename(rel)
            with open(os.path.join(REF, rel)) as f:
                src = f.read()
            with open(os.path.join(scratch, name), "w") as f:
                f.write(src.replace("cuda:0", "cpu"))
            p = subprocess.run([sys.executable, "-m", "pytest", name, "-q", "-p", "no:cacheprovider", "-rf"], cwd=scratch, env=env,
                               capture_output=True, text=True, timeout=1800)
            tail = [l for l in p.stdout.splitlines() if l.strip()]
            failed = [re.sub(r" - .*", "", l[len("FAILED "):]) for l in tail if l.startswith("FAILED ")]
            why = sorted({l.split(" - ", 1)[1] for l in tail if l.startswith("FAILED ") and " - " in l})
            report["files"][rel] = {"rc": p.returncode, "summary": tail[-1].strip("= ") if tail else "", "failed": failed, "failure_messages": why}
            print(rel, "->", report["files"][rel]["summary"])
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    report["note"] = ("test_voxel_collision.py::TestStandaloneComputeLocalSdf (6 tests) launches kernels defined in the test file itself "
                      "that call compute_local_sdf_with_grad(obs_set, env_idx, local_idx, local_pt) with four arguments, while "
                      "geom/data/data_voxel.py:1163 takes five (query_distance): a stale call in the reference's test (TypeError here, a "
                      "compile error under Warp), not reached by the kernels of the path")
    out = os.path.join(ROOT, "profiles", "r02_reference_warp_tests_on_cpu.json")
    with open(out, "w") as f:
        json.dump(report, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()
