from .lbfgs import LBFGSOpt, LBFGSOptCfg  # noqa: F401
from .mppi import MPPI, MPPICfg  # noqa: F401
from .pipelined import PipelinedLBFGS  # noqa: F401
from .multi_stage import MultiStageOptimizer  # noqa: F401
