"""The PRODUCT's MPPI particle sample library (curobo_amd/optim/particle_samples.py) against the reference's own
``MixedParticleSampler`` / ``GaussianDistribution.initialize_samples`` run on the CPU: random horizons, dimensions, seeds,
sample counts, halton / stomp ratios, filter coefficients.   python tests/randomised/sweep_reference_torch_samples.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/curobo/_src/optim/particle"):
    print("no /root/reference here: nothing to compare; 0 failed")
    sys.exit(0)
import torch  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")  # (`curobo` = the reference's package in this process; the product is `curobo_amd`)
from curobo._src.optim.components.gaussian_distribution import CovType, GaussianDistribution  # noqa: E402
from curobo._src.optim.particle.sample_strategies.particle_sampler import MixedParticleSampler  # noqa: E402
from curobo._src.optim.particle.sample_strategies.particle_sampler_cfg import ParticleSamplerCfg  # noqa: E402
from curobo._src.types.device_cfg import DeviceCfg  # noqa: E402

from curobo_amd.optim.particle_samples import ParticleSampleLib, sample_set  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dc = DeviceCfg(device=torch.device("cpu"))
bad = 0
for case in range(n_cases):
    H, D, seed, n = int(rng.choice([4, 8, 12, 16, 30])), int(rng.integers(1, 13)), int(rng.integers(0, 50)), int(rng.integers(1, 120))
    ratio = [{"halton": 1.0}, {"halton": 0.0, "stomp": 1.0}, {"halton": 0.5, "stomp": 0.5}, {"halton": 0.7, "stomp": 0.3}][int(rng.integers(4))]
    coeffs = [None, [0.3, 0.3, 0.4], [0.1, 0.2, 0.7]][int(rng.integers(3))]
    try:
        # fixed_samples = True (the optimiser's setting): the reference draws once and returns that set again; False: the
        # stream continues -- which is what ParticleSampleLib.get_samples does (the optimiser here draws its set once, sample_set)
        fixed = bool(rng.random() < 0.5)
        ref = MixedParticleSampler(ParticleSamplerCfg(device_cfg=dc, fixed_samples=fixed, sample_ratio=ratio, seed=seed, filter_coeffs=coeffs), H, D)
        ours = ParticleSampleLib(H, D, seed=seed, sample_ratio=ratio, filter_coeffs=None if coeffs is None else np.array(coeffs, np.float32))
        first = None
        for rnd in range(1 if fixed else 3):
            a, b = ref.get_samples([n]).numpy(), ours.get_samples(n).numpy()
            assert a.shape == b.shape, f"shape {a.shape} vs {b.shape}"
            np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-6, err_msg=f"samples (draw {rnd}, fixed_samples {fixed})")
            first = a if first is None else first
        if fixed:
            assert np.array_equal(ref.get_samples([n]).numpy(), first)
        if rng.random() < 0.5:
            P, m = int(rng.integers(1, 5)), int(rng.integers(2, 20))
            dist = GaussianDistribution(dc, H, D, CovType.DIAG_A, torch.zeros(1, H, D), torch.ones(1, D) * 0.5,
                                        ParticleSamplerCfg(device_cfg=dc, fixed_samples=True, seed=seed), seed=seed)
            dist.initialize_samples(P, m, 10, True, True)
            np.testing.assert_allclose(sample_set(ParticleSampleLib(H, D, seed=seed), P, m).numpy(), dist._sample_set.numpy(), rtol=1e-6, atol=1e-6,
                                       err_msg="the optimiser's pre-generated set")
    except AssertionError as e:
        bad += 1
        print(f"FAILED case {case}: H {H} D {D} seed {seed} n {n} ratio {ratio} coeffs {coeffs}: {str(e)[:300]}".replace("\n", " | "))
print(f"{n_cases} cases, {bad} failed")
