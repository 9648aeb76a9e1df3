"""forward kinematics front end (``curobo_amd.kinematics``)"""
from curobo_amd.kinematics import Kinematics, KinematicsCfg, KinematicsState  # noqa: F401

__all__ = ["Kinematics", "KinematicsCfg", "KinematicsState"]
