"""receding-horizon control (``curobo_amd.solver.mpc``; reference curobo/model_predictive_control.py)"""
from curobo_amd.solver.mpc import MPCSolver as ModelPredictiveControl  # noqa: F401
from curobo_amd.solver.mpc import MPCSolverCfg as ModelPredictiveControlCfg  # noqa: F401
from curobo_amd.solver.mpc import MPCSolverResult as ModelPredictiveControlResult  # noqa: F401

__all__ = ["ModelPredictiveControl", "ModelPredictiveControlCfg", "ModelPredictiveControlResult"]
