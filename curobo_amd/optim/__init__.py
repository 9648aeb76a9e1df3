from .lbfgs import LBFGSOpt, LBFGSOptCfg  # noqa: F401
from .mppi import MPPI, MPPICfg  # noqa: F401
