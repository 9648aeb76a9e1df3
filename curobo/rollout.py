"""rollouts: what an optimiser minimises (Rollout protocol, Rosenbrock test function, the robot rollouts)"""
from curobo_amd.rollout import (CollisionRollout, CollisionRolloutCfg, CostCollection, CostsAndConstraints, Rollout,  # noqa: F401
                                RolloutMetrics, RolloutResult, RosenbrockCfg, RosenbrockRollout, TrajOptRollout,
                                TrajOptRolloutCfg)

__all__ = ["RosenbrockCfg", "RosenbrockRollout", "Rollout", "RolloutResult", "RolloutMetrics", "CostsAndConstraints",
           "CostCollection", "CollisionRollout", "CollisionRolloutCfg", "TrajOptRollout", "TrajOptRolloutCfg"]
