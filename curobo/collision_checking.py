"""robot - world / self collision queries (``curobo_amd.collision_checking``)"""
from curobo_amd.collision_checking import RobotCollisionChecker, RobotCollisionCheckerCfg  # noqa: F401

__all__ = ["RobotCollisionChecker", "RobotCollisionCheckerCfg"]
