from .lbfgs import LBFGSOpt, LBFGSOptCfg  # noqa: F401
