"""``enable_tool_pose_tracking`` / ``disable_tool_pose_tracking`` of the reference's solvers (solver/solver_core.py:370-401; the
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
