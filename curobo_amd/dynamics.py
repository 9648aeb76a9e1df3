"""``Dynamics``: batched inverse dynamics front end, mirroring
``curobo/_src/robot/dynamics/dynamics.py:40-330`` (``setup_batch_size`` +
``compute_inverse_dynamics(position, velocity, acceleration, f_ext)`` with autograd)."""

from __future__ import annotations

from typing import Optional, Sequence

import torch

from .hip_ops.dynamics import RNEAForwardFunction
from .robot.kinematics_params import KinematicsParams


class Dynamics:
    def __init__(self, kinematics_config: KinematicsParams, gravity: Sequence[float] = (0.0, 0.0, -9.81)):
        kp = kinematics_config
        self.kinematics_config = kp
        self.dof = kp.num_dof
        g = torch.zeros(6, device=kp.device, dtype=torch.float32)
        # Featherstone: gravity enters as the base's spatial acceleration -g (reference
        # DynamicsCfg.get_gravity_spatial, robot/dynamics/dynamics_cfg.py)
        g[3:] = -torch.as_tensor(gravity, dtype=torch.float32)
        self._gravity_spatial = g
        self._total = 0
        self.setup_batch_size(1)

    def setup_batch_size(self, batch_size: int, horizon: int = 1) -> None:
        kp, n = self.kinematics_config, batch_size * horizon
        if n == self._total:
            return
        z = lambda *s: torch.zeros(*s, device=kp.device, dtype=torch.float32)  # noqa: E731
        self._total = n
        self._tau, self._gq, self._gqd, self._gqdd = z(n, self.dof), z(n, self.dof), z(n, self.dof), z(n, self.dof)
        self._cache = z(n, kp.num_links * 20)
        self._gfe = None

    def compute_inverse_dynamics(self, position: torch.Tensor, velocity: torch.Tensor, acceleration: torch.Tensor,
                                 f_ext: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tau with the shape of ``position`` ([dof], [batch, dof] or [batch, horizon, dof]);
        ``f_ext`` [batch, num_links, 6] or [num_links, 6] are subtracted link wrenches."""
        kp = self.kinematics_config
        shape = position.shape
        q = position.reshape(-1, self.dof).contiguous().float()
        qd = velocity.reshape(-1, self.dof).contiguous().float()
        qdd = acceleration.reshape(-1, self.dof).contiguous().float()
        n = q.shape[0]
        if n != self._total:
            self.setup_batch_size(n)
        fe, gfe = None, None
        if f_ext is not None:
            fe = (f_ext.unsqueeze(0).expand(n, -1, -1) if f_ext.dim() == 2 else f_ext.reshape(n, kp.num_links, 6)).contiguous()
            if fe.requires_grad:
                if self._gfe is None or self._gfe.shape[0] != n:
                    self._gfe = torch.zeros(n, kp.num_links, 6, device=kp.device)
                gfe = self._gfe
        tau = RNEAForwardFunction.apply(
            q, qd, qdd, self._tau, self._gq, self._gqd, self._gqdd, self._cache, kp.fixed_transforms,
            kp.link_masses_com, kp.link_inertias, kp.joint_map_type, kp.joint_map, kp.link_map, kp.joint_offset_map,
            self._gravity_spatial, kp.link_level_offsets, kp.link_level_data, kp.num_links, self.dof,
            kp.n_tree_levels, 1, fe, gfe)
        return tau.reshape(shape)
