"""IK -> trajectory optimisation -> time-optimal finetune (``curobo_amd.motion_planner``; reference curobo/motion_planner.py;
no graph planner, no grasp planning)"""
from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg  # noqa: F401

__all__ = ["MotionPlanner", "MotionPlannerCfg"]
