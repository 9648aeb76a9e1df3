"""many planning problems per IK + trajectory-optimisation pass (``curobo_amd.motion_planner``; reference
curobo/batch_motion_planner.py)"""
from curobo_amd.motion_planner import BatchMotionPlanner, MotionPlannerCfg  # noqa: F401

__all__ = ["BatchMotionPlanner", "MotionPlannerCfg"]
