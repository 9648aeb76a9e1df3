from .ik import IKResult, IKSolver, IKSolverCfg  # noqa: F401
from .trajopt import TrajOptResult, TrajOptSolver, TrajOptSolverCfg  # noqa: F401
from .seed_ik import SeedIKSolver, SeedIKSolverCfg  # noqa: F401
from .mpc import MPCSolver, MPCSolverCfg, MPCSolverResult  # noqa: F401
from .inverse_kinematics import InverseKinematics, InverseKinematicsCfg, InverseKinematicsResult  # noqa: F401
