"""The particle noise of the MPPI optimiser reproduces the reference's sample library: Halton
points through the 2000-row buffer, the erfinv map, the three-tap filter; STOMP-correlated noise; the pre-generated
per-problem sample set.  Golden: ``tests/golden/mppi_samples_golden.npz`` from the reference's ``MixedParticleSampler`` /
``GaussianDistribution`` run on CPU (``tests/golden/make_mppi_samples_golden.py``)."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN_DIR, "mppi_samples_golden.npz"))


@pytest.mark.parametrize("name", ["halton", "halton_nofilter", "stomp", "mixed"])
def test_sample_library_matches_the_reference(name, golden):
    from curobo_amd.optim.particle_samples import ParticleSampleLib

    H, D, seed, n = (int(v) for v in golden[f"{name}/params"])
    coeffs = golden[f"{name}/coeffs"]
    ratio = dict(zip([str(k) for k in golden[f"{name}/ratio_keys"]], [float(v) for v in golden[f"{name}/ratio_vals"]]))
    lib = ParticleSampleLib(H, D, seed=seed, sample_ratio=ratio, filter_coeffs=None if coeffs.size == 0 else coeffs)
    got = lib.get_samples(n).numpy()
    ref = golden[f"{name}/samples"]
    assert got.shape == ref.shape
    # same scipy Halton points, same CPU index stream, same torch arithmetic: bit-identical in the build container; a few
    # ulp of erfinv / matmul are allowed for another torch build
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
    # the stream continues (no reuse) and rewinds
    again = lib.get_samples(n).numpy()
    assert not np.array_equal(again, got)
    lib.reset_seed()
    np.testing.assert_array_equal(lib.get_samples(n).numpy(), got)


def test_sample_set_of_an_optimiser_matches_the_reference(golden):
    from curobo_amd.optim.particle_samples import ParticleSampleLib, sample_set

    H, D, seed, P, n = (int(v) for v in golden["set/params"])
    s = sample_set(ParticleSampleLib(H, D, seed=seed), P, n).numpy()
    np.testing.assert_allclose(s, golden["set/samples"], rtol=1e-6, atol=1e-6)
    assert (s[:, :, -1] == 0).all() and np.abs(s[:, :, :-1]).max() > 0.1


def test_stomp_noise_is_smooth_and_pinned_at_the_ends():
    from curobo_amd.optim.particle_samples import ParticleSampleLib

    x = ParticleSampleLib(24, 5, seed=1, sample_ratio={"stomp": 1.0}).get_samples(50).numpy()
    assert (x[:, 0] == 0).all() and (x[:, -2:] == 0).all() and np.abs(x).max() == pytest.approx(1.0)
    rough = np.abs(np.diff(x[:, 1:-2], n=2, axis=1)).mean()
    white = np.abs(np.diff(np.random.default_rng(0).normal(size=x[:, 1:-2].shape) * x.std(), n=2, axis=1)).mean()
    assert rough < 0.3 * white


@pytest.mark.skipif(not os.path.isdir("/root/reference/curobo/_src/optim/particle"), reason="the reference's sample library is not on this machine")
def test_randomised_sweep_against_the_reference_sample_library():
    """tests/randomised/sweep_reference_torch_samples.py at a small size: random horizons, dimensions, seeds, counts, ratios, filters;
    continuing streams (fixed_samples off) and the optimiser's pre-generated set"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "randomised", "sweep_reference_torch_samples.py"), "40", "9"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, (out.stdout + out.stderr)[-2000:]
