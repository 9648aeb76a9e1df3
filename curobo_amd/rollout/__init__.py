from .collision_rollout import CollisionRollout, CollisionRolloutCfg  # noqa: F401
from .trajopt_rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: F401
from .protocol import CostCollection, CostsAndConstraints, Rollout, RolloutMetrics, RolloutResult  # noqa: F401
from .rosenbrock import RosenbrockCfg, RosenbrockRollout  # noqa: F401
