from .ik import IKResult, IKSolver, IKSolverCfg  # noqa: F401
from .trajopt import TrajOptResult, TrajOptSolver, TrajOptSolverCfg  # noqa: F401
from .seed_ik import SeedIKSolver, SeedIKSolverCfg  # noqa: F401
