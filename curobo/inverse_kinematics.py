"""collision-aware batched inverse kinematics (``curobo_amd.solver.inverse_kinematics``)"""
from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg, InverseKinematicsResult  # noqa: F401

__all__ = ["InverseKinematics", "InverseKinematicsCfg", "InverseKinematicsResult"]
