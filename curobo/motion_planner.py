"""IK -> trajectory optimisation -> time-optimal finetune, grasp planning (``curobo_amd.motion_planner``; reference
curobo/motion_planner.py; no graph planner)"""
from curobo_amd.motion_planner import GraspPlanResult, MotionPlanner, MotionPlannerCfg  # noqa: F401

__all__ = ["MotionPlanner", "MotionPlannerCfg", "GraspPlanResult"]
