"""Members every solver front end of the reference has through its solver core: ``sample_configs`` (solver_core.py:447-480) and
``enable_tool_pose_tracking`` / ``disable_tool_pose_tracking`` of the reference's solvers (solver/solver_core.py:370-401; the
MPC front end forwards its configured non-terminal factor, solver_mpc.py:190-197): switch the pose cost of some (or all) tool
frames on with the standard position + orientation criteria, or off, through ``update_tool_pose_criteria``.  A class that mixes
this in has ``tool_frames`` and ``update_tool_pose_criteria``; criteria of the frames that are not named stay as they are."""

from __future__ import annotations

from typing import Dict, List, Optional

from ..types import ToolPoseCriteria


class ToolPoseTrackingMixin:
    #: non-terminal factor ``enable_tool_pose_tracking`` uses when the caller gives none (the reference's solver core: 0.0)
    _tracking_non_terminal_factor: float = 0.0

    def _merged_criteria(self, update: Dict[str, ToolPoseCriteria]) -> Dict[str, ToolPoseCriteria]:
        cur = dict(getattr(self, "_criteria", None) or {})
        cur.update(update)
        return cur

    def enable_tool_pose_tracking(self, tool_frames: Optional[List[str]] = None, non_terminal_weight_factor: Optional[float] = None) -> None:
        frames = list(self.tool_frames) if tool_frames is None else list(tool_frames)
        f = self._tracking_non_terminal_factor if non_terminal_weight_factor is None else float(non_terminal_weight_factor)
        self.update_tool_pose_criteria(self._merged_criteria(
            {k: ToolPoseCriteria.track_position_and_orientation(non_terminal_scale=f) for k in frames}))

    def disable_tool_pose_tracking(self, tool_frames: Optional[List[str]] = None) -> None:
        frames = list(self.tool_frames) if tool_frames is None else list(tool_frames)
        self.update_tool_pose_criteria(self._merged_criteria({k: ToolPoseCriteria.disabled() for k in frames}))

    def sample_configs(self, num_samples: int, rejection_ratio: int = 10) -> "torch.Tensor":  # noqa: F821
        """collision-free joint configurations [<= num_samples, dof] inside the joint limits (reference sample_configs): rejection
        sampling through a ``RobotCollisionChecker`` over the front end's robot and CURRENT scene"""
        from ..collision_checking import RobotCollisionChecker

        if num_samples <= 0:
            raise ValueError("num_samples must be positive")
        scene = getattr(self.config, "scene", None)
        cached = getattr(self, "_sample_checker", None)
        if cached is None or cached[0] is not scene:
            cached = (scene, RobotCollisionChecker(self.config.kinematics, scene))
            self._sample_checker = cached
        chk = cached[1]
        chk.rejection_ratio = rejection_ratio
        return chk.sample(num_samples, mask_valid=True)

    @property
    def attachment_manager(self):
        """attaching / detaching obstacles to a robot link (reference solver core :81-86; the planners hand out the trajectory
        optimiser's, motion_planner.py:103-106): ONE manager per front end over its ``Kinematics`` -- whose ``link_spheres`` every
        solver of the front end reads -- and its CURRENT scene"""
        from ..attachment_manager import AttachmentManager

        owner = getattr(self, "trajopt_solver", self)
        scene = getattr(owner.config, "scene", None)
        m = getattr(owner, "_attachment_manager", None)
        if m is None:
            m = owner._attachment_manager = AttachmentManager(owner.kinematics, scene)
        m.update_world(scene)
        return m
