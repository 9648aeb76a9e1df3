"""``curobo`` namespace of the MI355X-native implementation: the reference's public module names over
``curobo_amd`` (HIP kernels, no CUDA / Warp).  ``from curobo.kinematics import Kinematics, KinematicsCfg``,
``from curobo.collision_checking import RobotCollisionChecker``, ``from curobo.optim import LBFGSOpt``,
``from curobo.rollout import RosenbrockRollout``, ``from curobo.inverse_kinematics import InverseKinematics``,
``from curobo.trajectory_optimizer import TrajectoryOptimizer``, ``from curobo.motion_planner import MotionPlanner``,
``from curobo.batch_motion_planner import BatchMotionPlanner`` and
``from curobo.types import JointState`` resolve to the classes documented in ``curobo_amd`` (each cites the reference
file it mirrors).  Only the motion-generation hot path is covered: graph search, perception, viewers are out of scope."""

from curobo_amd import __version__  # noqa: F401
from curobo_amd.motion_planner import (BatchMotionPlanner, MotionPlanner, MotionPlannerCfg,  # noqa: F401
                                       TrajectoryOptimizer, TrajectoryOptimizerCfg)
from curobo_amd.kinematics import Kinematics, KinematicsCfg  # noqa: F401
from curobo_amd.model_predictive_control import ModelPredictiveControl, ModelPredictiveControlCfg  # noqa: F401
from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg  # noqa: F401

__all__ = ["InverseKinematics", "InverseKinematicsCfg", "TrajectoryOptimizer", "TrajectoryOptimizerCfg", "MotionPlanner",
           "MotionPlannerCfg", "BatchMotionPlanner", "ModelPredictiveControl", "ModelPredictiveControlCfg", "Kinematics", "KinematicsCfg"]
