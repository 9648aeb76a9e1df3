"""optimisers with the reference's call surface: ``LBFGSOpt(config, rollout_list, use_cuda_graph)`` etc."""
from curobo_amd.optim.mppi import MPPI, MPPICfg  # noqa: F401
from curobo_amd.optim.multi_stage import MultiStageOptimizer  # noqa: F401
from curobo_amd.optim.reference_api import LBFGSOpt, LBFGSOptCfg  # noqa: F401

__all__ = ["LBFGSOpt", "LBFGSOptCfg", "MPPI", "MPPICfg", "MultiStageOptimizer"]
