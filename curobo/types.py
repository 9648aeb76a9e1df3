"""common data types (``curobo_amd.types``)"""
from curobo_amd.kinematics import ToolPose  # noqa: F401
from curobo_amd.types import DeviceCfg, GoalToolPose, JointState, Pose, ToolPoseCriteria  # noqa: F401

__all__ = ["JointState", "Pose", "ToolPose", "GoalToolPose", "ToolPoseCriteria", "DeviceCfg"]
