"""receding-horizon control (``curobo_amd.model_predictive_control`` over ``curobo_amd.solver.mpc``; reference
curobo/model_predictive_control.py)"""
from curobo_amd.model_predictive_control import (ModelPredictiveControl, ModelPredictiveControlCfg,  # noqa: F401
                                                 ModelPredictiveControlResult)

__all__ = ["ModelPredictiveControl", "ModelPredictiveControlCfg", "ModelPredictiveControlResult"]
