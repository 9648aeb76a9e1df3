"""Golden particle sets from the REFERENCE's sample library, run on CPU in the build container:
``MixedParticleSampler`` (halton + the three-tap filter; stomp) and ``GaussianDistribution.initialize_samples``.
    PYTHONPATH=/root/reference python tests/golden/make_mppi_samples_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from curobo._src.optim.components.gaussian_distribution import CovType, GaussianDistribution
    from curobo._src.optim.particle.sample_strategies.particle_sampler import MixedParticleSampler
    from curobo._src.optim.particle.sample_strategies.particle_sampler_cfg import ParticleSamplerCfg
    from curobo._src.types.device_cfg import DeviceCfg

    dc = DeviceCfg(device=torch.device("cpu"))
    out = {}
    cases = [("halton", {"halton": 1.0}, 12, 7, 3, 40, [0.3, 0.3, 0.4]),
             ("halton_nofilter", {"halton": 1.0}, 8, 6, 11, 25, None),
             ("stomp", {"halton": 0.0, "stomp": 1.0}, 16, 7, 5, 30, [0.3, 0.3, 0.4]),
             ("mixed", {"halton": 0.5, "stomp": 0.5}, 12, 7, 2, 40, [0.3, 0.3, 0.4])]
    for name, ratio, H, D, seed, n, coeffs in cases:
        cfg = ParticleSamplerCfg(device_cfg=dc, fixed_samples=True, sample_ratio=ratio, seed=seed, filter_coeffs=coeffs)
        lib = MixedParticleSampler(cfg, H, D)
        out[f"{name}/samples"] = lib.get_samples([n]).numpy().copy()
        out[f"{name}/params"] = np.array([H, D, seed, n], np.int64)
        out[f"{name}/coeffs"] = np.array(coeffs if coeffs is not None else [], np.float32)
        out[f"{name}/ratio_keys"] = np.array(list(ratio.keys()))
        out[f"{name}/ratio_vals"] = np.array(list(ratio.values()), np.float64)
    # the optimiser's pre-generated set: 3 problems x 9 sampled particles, fixed samples, per problem
    H, D, seed, P, n = 12, 7, 4, 3, 9
    cfg = ParticleSamplerCfg(device_cfg=dc, fixed_samples=True, seed=seed)
    dist = GaussianDistribution(dc, H, D, CovType.DIAG_A, torch.zeros(1, H, D), torch.ones(1, D) * 0.5, cfg, seed=seed)
    dist.initialize_samples(P, n, 10, True, True)
    out["set/samples"] = dist._sample_set.numpy().copy()
    out["set/params"] = np.array([H, D, seed, P, n], np.int64)
    np.savez_compressed(os.path.join(HERE, "mppi_samples_golden.npz"), **out)
    print("wrote mppi_samples_golden.npz", {k: v.shape for k, v in out.items() if k.endswith("samples")})


if __name__ == "__main__":
    main()
