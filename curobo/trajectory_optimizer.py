"""collision-aware trajectory optimisation (``curobo_amd.motion_planner``; reference curobo/trajectory_optimizer.py)"""
from curobo_amd.motion_planner import TrajectoryOptimizer, TrajectoryOptimizerCfg, TrajectoryOptimizerResult  # noqa: F401

__all__ = ["TrajectoryOptimizer", "TrajectoryOptimizerCfg", "TrajectoryOptimizerResult"]
