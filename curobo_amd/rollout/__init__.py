from .collision_rollout import CollisionRollout, CollisionRolloutCfg  # noqa: F401
from .trajopt_rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: F401
