"""``KinematicsFusedFunction`` -- reference ``cuda_ops/kinematics.py:25-378``."""

from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from ..backends import kinematics as kinematics_hip
from .tensor_checks import (
    check_bool_tensors,
    check_float32_tensors,
    check_int8_tensors,
    check_int16_tensors,
    check_int32_tensors,
)


class KinematicsFusedFunction(Function):
    @staticmethod
    def create_buffers(batch: int, horizon: int, kinematics_config, device=None):
        """reference :27-90"""
        k = kinematics_config
        device = device or k.device
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)  # noqa: E731
        buffers = {
            "batch_link_position": z(batch, horizon, k.num_pose_links, 3),
            "batch_link_quaternion": z(batch, horizon, k.num_pose_links, 4),
            "batch_robot_spheres": z(batch, horizon, k.num_spheres, 4),
            "batch_com": z(batch, horizon, 4),
            "batch_jacobian": z(batch, horizon, k.num_pose_links, 6, k.num_dof),
            "batch_cumul_mat": z(batch, horizon, k.num_links, 3, 4),
            "grad_out_q": z(batch, horizon, k.num_dof),
            "grad_out_q_jacobian": z(batch, horizon, k.num_dof),
        }
        buffers["grad_in_link_pos"] = torch.zeros_like(buffers["batch_link_position"])
        buffers["grad_in_link_quat"] = torch.zeros_like(buffers["batch_link_quaternion"])
        buffers["grad_in_robot_spheres"] = torch.zeros_like(buffers["batch_robot_spheres"])
        buffers["grad_in_com"] = torch.zeros_like(buffers["batch_com"])
        return buffers

    @staticmethod
    def forward(ctx, joint_seq, batch_link_position, batch_link_quaternion, batch_robot_spheres, batch_com,
                batch_jacobian, batch_cumul_mat, kinematics_config, grad_out, grad_out_q_jacobian,
                grad_in_link_pos, grad_in_link_quat, grad_in_robot_spheres, grad_in_com, compute_jacobian: bool,
                compute_spheres: bool, compute_com: bool, env_query_idx, horizon: int):
        k = kinematics_config
        b_size = batch_link_position.shape[0] * batch_link_position.shape[1]
        num_spheres = batch_robot_spheres.shape[2] if compute_spheres else 0
        n_joints = joint_seq.shape[-1]
        ctx.set_materialize_grads(False)
        device = joint_seq.device
        k.validate_shapes()  # reference :155
        check_float32_tensors(
            device, joint_seq=joint_seq, batch_link_position=batch_link_position,
            batch_link_quaternion=batch_link_quaternion, batch_robot_spheres=batch_robot_spheres,
            batch_com=batch_com, batch_jacobian=batch_jacobian, batch_cumul_mat=batch_cumul_mat,
            fixed_transforms=k.fixed_transforms, link_spheres=k.link_spheres,
            link_masses_com=k.link_masses_com, joint_offset_map=k.joint_offset_map)
        check_int16_tensors(
            device, link_map=k.link_map, joint_map=k.joint_map, tool_frame_map=k.tool_frame_map,
            link_sphere_idx_map=k.link_sphere_idx_map, link_chain_data=k.link_chain_data,
            link_chain_offsets=k.link_chain_offsets, joint_links_data=k.joint_links_data,
            joint_links_offsets=k.joint_links_offsets)
        check_int8_tensors(device, joint_map_type=k.joint_map_type)
        check_bool_tensors(device, joint_affects_endeffector=k.joint_affects_endeffector)
        check_int32_tensors(device, env_query_idx=env_query_idx)
        if compute_jacobian:
            kinematics_hip.launch_kinematics_forward_spheres_jacobian(
                batch_link_position, batch_link_quaternion, batch_robot_spheres, batch_com, batch_jacobian,
                batch_cumul_mat, joint_seq, k.fixed_transforms, k.link_spheres, k.link_masses_com,
                k.joint_map_type, k.joint_map, k.link_map, k.tool_frame_map, k.link_sphere_idx_map,
                k.link_chain_data, k.link_chain_offsets, k.joint_links_data, k.joint_links_offsets,
                k.joint_affects_endeffector, k.joint_offset_map, env_query_idx, k.num_envs, b_size, horizon,
                n_joints, num_spheres, 32, write_global_cumul=True, compute_com=compute_com)
        elif num_spheres > 0:
            kinematics_hip.launch_kinematics_forward_spheres(
                batch_link_position, batch_link_quaternion, batch_robot_spheres, batch_com, batch_cumul_mat,
                joint_seq, k.fixed_transforms, k.link_spheres, k.link_masses_com, k.joint_map_type, k.joint_map,
                k.link_map, k.tool_frame_map, k.link_sphere_idx_map, k.joint_offset_map, env_query_idx,
                k.num_envs, b_size, horizon, n_joints, num_spheres, 32, write_global_cumul=True,
                compute_com=compute_com)
        else:
            kinematics_hip.launch_kinematics_forward(
                batch_link_position, batch_link_quaternion, batch_com, batch_cumul_mat, joint_seq,
                k.fixed_transforms, k.link_masses_com, k.joint_map_type, k.joint_map, k.link_map,
                k.tool_frame_map, k.joint_offset_map, b_size, horizon, n_joints, compute_com)
        ctx.mark_non_differentiable(batch_cumul_mat, grad_out)
        ctx.kinematics_config = k
        ctx.compute_jacobian, ctx.compute_spheres, ctx.compute_com = compute_jacobian, compute_spheres, compute_com
        ctx.env_query_idx, ctx.horizon = env_query_idx, horizon
        ctx.grad_in_link_pos, ctx.grad_in_link_quat = grad_in_link_pos, grad_in_link_quat
        ctx.grad_in_robot_spheres, ctx.grad_in_com = grad_in_robot_spheres, grad_in_com
        ctx.save_for_backward(joint_seq, grad_out, batch_cumul_mat, batch_com)
        return batch_link_position, batch_link_quaternion, batch_robot_spheres, batch_com, batch_jacobian

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_in_link_pos: Optional[torch.Tensor], grad_in_link_quat: Optional[torch.Tensor],
                 grad_in_spheres: Optional[torch.Tensor], grad_in_com: Optional[torch.Tensor],
                 grad_in_link_jacobian: Optional[torch.Tensor]):
        grad_joint = None
        if ctx.needs_input_grad[0]:
            joint_seq, grad_out, batch_cumul_mat, batch_com = ctx.saved_tensors
            k = ctx.kinematics_config
            # None grads (set_materialize_grads(False)) are replaced by the zero buffers, :287-294
            grad_in_link_pos = ctx.grad_in_link_pos if grad_in_link_pos is None else grad_in_link_pos
            grad_in_link_quat = ctx.grad_in_link_quat if grad_in_link_quat is None else grad_in_link_quat
            grad_in_spheres = ctx.grad_in_robot_spheres if grad_in_spheres is None else grad_in_spheres
            grad_in_com = ctx.grad_in_com if grad_in_com is None else grad_in_com
            num_spheres = grad_in_spheres.shape[2] if ctx.compute_spheres else 0
            check_float32_tensors(
                joint_seq.device, grad_in_spheres=grad_in_spheres, grad_out=grad_out,
                grad_in_link_pos=grad_in_link_pos, grad_in_link_quat=grad_in_link_quat,
                batch_cumul_mat=batch_cumul_mat, batch_com=batch_com, grad_in_com=grad_in_com)
            if grad_in_link_quat.data_ptr() % 16 != 0:
                raise ValueError("grad_in_link_quat is not aligned to 16 bytes")
            jac_grad = ctx.compute_jacobian and grad_in_link_jacobian is not None  # dJ/dq (reference JAC_GRAD, :268-378)
            if jac_grad:
                grad_in_link_jacobian = grad_in_link_jacobian.contiguous()
                check_float32_tensors(joint_seq.device, grad_in_link_jacobian=grad_in_link_jacobian)
            b_size = joint_seq.shape[0] * joint_seq.shape[1]
            kinematics_hip.launch_kinematics_backward(
                grad_out, grad_in_link_pos, grad_in_link_quat, grad_in_spheres, grad_in_com, batch_com,
                grad_in_link_jacobian if jac_grad else grad_in_link_pos, batch_cumul_mat, k.link_spheres, k.link_masses_com, k.link_map, k.joint_map,
                k.joint_map_type, k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data,
                k.link_chain_offsets, k.joint_links_data, k.joint_links_offsets, k.joint_affects_endeffector,
                k.joint_offset_map, ctx.env_query_idx, k.num_envs, b_size, ctx.horizon, joint_seq.shape[-1],
                num_spheres, ctx.compute_com, jac_grad)
            grad_joint = grad_out
        return (grad_joint,) + (None,) * 18
