from .collision_rollout import CollisionRollout, CollisionRolloutCfg  # noqa: F401
