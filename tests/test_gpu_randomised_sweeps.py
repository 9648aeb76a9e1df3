"""The randomised parity sweeps of tests/randomised/ (random worlds, shapes and degenerate inputs; HIP against the oracle, the fused
launches against the kernel sequence) at a size that fits the suite.  Longer runs: ``python tests/randomised/fuzz_*.py <cases> <seed>`` (fuzz_self.py: self collision at 127 k configurations)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args,ok", [
    ("fuzz_fused.py", ["24", "5"], ", 0 failed"),
    ("fuzz_trajopt.py", ["16", "5"], ", 0 failed"),
    ("fuzz_scene.py", ["24", "5"], ", 0 failed"),
    ("fuzz_fk_bspline.py", ["16", "5"], "failed in total: 0"),
    ("fuzz_rnea.py", ["6", "5"], "failed in total: 0"),
    ("fuzz_mesh.py", ["10", "5"], ", 0 failed"),
    ("fuzz_mesh.py", ["10", "6", "--open"], ", 0 failed"),  # open / flipped meshes against the reference's ray sign
    ("fuzz_planner.py", ["3", "5"], ", 0 failed"),
    ("fuzz_ik.py", ["5", "5"], ", 0 failed"),
    ("fuzz_costs.py", ["16", "5"], ", 0 failed"),
    ("fuzz_lm.py", ["30", "5"], ", 0 failed"),
    ("fuzz_mppi.py", ["30", "5"], ", 0 failed"),
    ("fuzz_opt.py", ["16", "5"], None),  # (its L-BFGS cases include ill-conditioned histories: the line-search half must be exact)
])
def test_randomised_sweep(script, args, ok):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "randomised", script), *args], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    if ok is not None:
        assert ok in text, text[-2000:]
    else:
        assert "line search FAILED" not in text and "line search cases" in text, text[-2000:]
