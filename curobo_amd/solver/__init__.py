from .ik import IKResult, IKSolver, IKSolverCfg  # noqa: F401
