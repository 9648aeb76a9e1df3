"""HIP kernel backend -- the MI355X counterpart of ``curobo._src.curobolib.backends``.

The reference selects a backend dict ``{kinematics, optimization, trajectory, geometry, dynamics,
pba}`` in ``curobo/_src/curobolib/backends/__init__.py:162-227``; each value is a module of
launch functions that take pre-allocated ``torch.Tensor`` s and mutate them in place.  The
modules here export the same function names with the same positional argument order, so
``get_backend()`` can return them unchanged.  ``collision`` is new: the
reference runs scene collision through NVIDIA Warp and has no backend hook for it.
"""

from . import collision, cost, dynamics, geometry, kinematics, linalg, optimization, rollout, trajectory  # noqa: F401


def get_backend():
    """Same shape as the reference's ``get_backend()`` result."""
    return {
        "kinematics": kinematics,
        "optimization": optimization,
        "trajectory": trajectory,
        "geometry": geometry,
        "collision": collision,
        "cost": cost,
        "rollout": rollout,
        "dynamics": dynamics,
        "linalg": linalg,
    }
