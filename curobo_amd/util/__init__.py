"""Host-side helpers around the hot path (pure torch): trajectory retiming / re-interpolation
(``trajectory``) and B-spline knot seeds (``knot_seeds``)."""

from .knot_seeds import TrajectorySeedGenerator
from .trajectory import calculate_dt_no_clamp, calculate_traj_steps, interpolate_bspline_knots

__all__ = ["TrajectorySeedGenerator", "calculate_dt_no_clamp", "calculate_traj_steps", "interpolate_bspline_knots"]
