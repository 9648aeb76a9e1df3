"""``torch.autograd.Function`` wrappers over the HIP backend -- the drop-in counterpart of the
reference's ``curobo/_src/curobolib/cuda_ops`` (FK, self collision, B-spline, L-BFGS, line
search) and of its Warp autograd functions for scene collision
(``curobo/_src/geom/collision/wp_autograd.py``).  Same class names, same ``forward`` argument
order, same autograd contract (pre-allocated buffers, ``once_differentiable``, saved
``batch_cumul_mat``, collision backward returns the buffer written by forward)."""

from .collision import CollisionBuffer, SphereObstacleCollision, SweptSphereObstacleCollision  # noqa: F401
from .geometry import SelfCollisionDistance  # noqa: F401
from .kinematics import KinematicsFusedFunction  # noqa: F401
from .optimization import LBFGScu, wolfe_line_search  # noqa: F401
from .trajectory import BSplineIdxKernel  # noqa: F401
from .cost import L2DistFunction, StateCSpaceFunction, ToolPoseDistance  # noqa: F401
