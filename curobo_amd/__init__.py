"""curobo_amd -- MI355X-native (HIP / gfx950) implementation of cuRobo's batched motion-generation
hot path: FK over the URDF tree, sphere-world + self-collision signed distance, the rollout
cost/gradient stack and the L-BFGS step, behind the reference's kernel-backend boundary
(``curobo._src.curobolib.backends``).  See DESIGN.md and INTEGRATION.md.
"""

__version__ = "0.1.0"
