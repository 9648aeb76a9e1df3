"""Cost terms of the robot rollout with the reference's class names and ``forward`` contracts
(``curobo/_src/cost``): thin modules over the HIP autograd functions (pre-allocated buffers,
``setup_batch_tensors``, weights as device tensors so they can change under a captured graph)."""

from .collision import SceneCollisionCost, SceneCollisionCostCfg, SelfCollisionCost, SelfCollisionCostCfg  # noqa: F401
from .tool_pose import ToolPoseCost, ToolPoseCostCfg  # noqa: F401
