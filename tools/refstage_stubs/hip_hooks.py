"""The Warp-side replacements of INTEGRATION.md applied to the (staged, otherwise unmodified) reference at run time:
the ``forward`` bodies of the reference's Warp-launching autograd functions become calls into libcurobo_hip.so through
``curobo_amd.backends``.  TEST INFRASTRUCTURE of tools/reference_on_hip.py (the reference's own callers and tests then run
on the MI355X with no NVIDIA Warp): a maintainer would paste these bodies into the files named next to each hook.

    geom/collision/wp_autograd.py:37-249          SphereObstacleCollision / SweptSphereObstacleCollision  -> collision backend
    cost/wp_tool_pose.py:698-914                  ToolPoseDistance                                         -> cost backend
    cost/wp_cspace_position.py:18-232             PositionCSpaceFunction                                   -> cost backend
    cost/wp_cspace_state.py:288-680               StateCSpaceFunction                                      -> cost backend
    cost/wp_torch_cspace_dist.py:81-158           L2DistFunction                                           -> cost backend
    optim/util/levenberg_marquardt_step.py:96-143 LevenbergMarquardtStep.forward                           -> linalg backend

Every hook counts its calls in ``CALLS`` (the report shows which entry points the reference's callers reached)."""
import sys

import torch

CALLS = {}


def _count(name):
    CALLS[name] = CALLS.get(name, 0) + 1


# ----------------------------------------------------------------------------- scene collision
_scene_cache = {}


def _scene_struct(scene):
    """``curobo_hip_scene`` of a reference ``SceneData`` (geom/data/data_scene.py:35-70): plain pointers of its cuboid and
    voxel stores; rebuilt when a store's tensor is replaced (``update_world``)"""
    from curobo_amd.backends import collision as collision_hip

    c, v = scene.cuboids, scene.voxels
    key = (id(scene), None if c is None else c.dims.data_ptr(), None if v is None else v.features.data_ptr())
    hit = _scene_cache.get(id(scene))
    if hit is not None and hit[0] == key:
        return hit[1]
    m = scene.meshes
    if m is not None and getattr(m, "count", None) is not None and int(torch.as_tensor(m.count).sum()) > 0:
        raise NotImplementedError("mesh obstacles: bake them into an ESDF grid (curobo_hip_mesh_esdf_bake)")
    kw = {}
    if c is not None:
        kw.update(cuboid_dims=c.dims, cuboid_inv_pose=c.inv_pose, cuboid_enable=c.enable, cuboid_count=c.count)
    if v is not None:
        kw.update(voxel_params=v.params, voxel_inv_pose=v.inv_pose, voxel_enable=v.enable, voxel_count=v.count,
                  voxel_features=v.features, voxel_max_distance=float(v.max_esdf_distance))
    s = collision_hip.make_scene(**kw)
    _scene_cache[id(scene)] = (key, s, scene)
    return s


def _collision_forward(ctx, query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env, return_loss,
                       sweep, speed_dt=None, enable_speed_metric=False):
    from curobo_amd.backends import collision as collision_hip

    b, h, n, _ = query_spheres.shape
    collision_hip.sphere_obstacle_collision(
        buffer.distance, buffer.gradient, query_spheres.detach().contiguous(), _scene_struct(scene), weight, activation_distance,
        env_query_idx.view(-1), b, h, n, bool(use_multi_env), 3 if sweep else 0, bool(enable_speed_metric), speed_dt)
    ctx.return_loss = return_loss
    ctx.save_for_backward(buffer.gradient)
    return buffer.distance


def sphere_forward(ctx, query_spheres, buffer, scene, weight, activation_distance, max_distance, env_query_idx, use_multi_env,
                   return_loss=False):
    _count("curobo_hip_sphere_obstacle_collision (SphereObstacleCollision.forward)")
    return _collision_forward(ctx, query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env,
                              return_loss, False)


def swept_forward(ctx, query_spheres, buffer, scene, weight, activation_distance, max_distance, speed_dt, enable_speed_metric,
                  env_query_idx, use_multi_env, return_loss=False):
    _count("curobo_hip_sphere_obstacle_collision, swept (SweptSphereObstacleCollision.forward)")
    return _collision_forward(ctx, query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env,
                              return_loss, True, speed_dt, enable_speed_metric)


# ----------------------------------------------------------------------------- tool pose
def tool_pose_forward(ctx, current_position, current_quat, goal_position, goal_quat, idxs_goal, position_orientation_weight,
                      terminal_pose_axes_weight_factor, non_terminal_pose_axes_weight_factor, terminal_pose_convergence_tolerance,
                      non_terminal_pose_convergence_tolerance, project_distance_to_goal, out_distance, out_position_distance,
                      out_rotation_distance, out_position_gradient, out_rotation_gradient, out_goalset_idx, use_grad_input, warp_kernel):
    from curobo_amd.backends import cost as cost_hip

    _count("curobo_hip_tool_pose_distance (ToolPoseDistance.forward)")
    ctx.set_materialize_grads(False)
    b, h, num_links, _ = current_position.shape
    num_goalset = goal_position.shape[-2]
    # the kernel handle names its compile-time constants: goalset_pose_distance_<num_goalset>_<rotation_method> (wp_tool_pose.py:694)
    rotation_method = int(str(getattr(warp_kernel, "__name__", "x_0")).rsplit("_", 1)[-1])
    cost_hip.tool_pose_distance(
        out_distance, out_position_distance, out_rotation_distance, out_position_gradient, out_rotation_gradient, out_goalset_idx,
        current_position.detach().contiguous(), current_quat.detach().contiguous(), goal_position.detach().contiguous(),
        goal_quat.detach().contiguous(), idxs_goal.detach().view(-1).contiguous(), position_orientation_weight,
        terminal_pose_axes_weight_factor, non_terminal_pose_axes_weight_factor, terminal_pose_convergence_tolerance,
        non_terminal_pose_convergence_tolerance, project_distance_to_goal, b, h, num_links, num_goalset, rotation_method)
    ctx.use_grad_input = use_grad_input
    ctx.mark_non_differentiable(out_position_distance, out_rotation_distance, out_goalset_idx, goal_position, goal_quat, idxs_goal,
                                position_orientation_weight, terminal_pose_axes_weight_factor, non_terminal_pose_axes_weight_factor,
                                terminal_pose_convergence_tolerance, non_terminal_pose_convergence_tolerance, project_distance_to_goal)
    ctx.save_for_backward(out_position_gradient, out_rotation_gradient)
    return out_distance, out_position_distance, out_rotation_distance, out_goalset_idx


# ----------------------------------------------------------------------------- c-space costs
def cspace_position_forward(ctx, pos, joint_torque, target_joint_position, idxs_target_joint_position, p_l, effort_limit, weight,
                            activation_distance, cspace_target_weight, cspace_target_dof_weight, squared_l2_regularization_weight,
                            current_position, current_velocity, idxs_current_state, v_b, state_dt, out_cost, out_gp, out_gtau,
                            use_grad_input):
    from curobo_amd.backends import cost as cost_hip

    _count("curobo_hip_cspace_position_cost (PositionCSpaceFunction.forward)")
    ctx.set_materialize_grads(False)
    b, h, dof = pos.shape
    if idxs_target_joint_position.ndim == 2:
        idxs_target_joint_position = idxs_target_joint_position.squeeze(1)
    if idxs_current_state.ndim == 2:
        idxs_current_state = idxs_current_state.squeeze(1)
    cost_hip.cspace_position_cost(
        out_cost, out_gp, out_gtau, pos.detach().contiguous(), joint_torque.detach().contiguous(), target_joint_position.detach(),
        idxs_target_joint_position.detach().contiguous(), p_l, effort_limit, weight, activation_distance, cspace_target_weight,
        cspace_target_dof_weight, squared_l2_regularization_weight, current_position.detach(), current_velocity.detach(),
        idxs_current_state.detach().contiguous(), v_b, state_dt.detach(), bool(pos.requires_grad), b, h, dof)
    ctx.use_grad_input = use_grad_input
    ctx.save_for_backward(out_gp, out_gtau)
    return out_cost


def cspace_state_forward(ctx, pos, vel, acc, jerk, joint_torque, state_dt, target_joint_position, idxs_target_joint_position, p_b, v_b,
                         a_b, j_b, effort_limit, weight, activation_distance, squared_l2_regularization_weights, cspace_target_weight,
                         cspace_non_terminal_weight_factor, cspace_target_dof_weight, out_cost, out_gp, out_gv, out_ga, out_gj, out_gtau,
                         retime_weights, retime_regularization_weights, use_grad_input):
    from curobo_amd.backends import cost as cost_hip

    _count("curobo_hip_cspace_state_cost (StateCSpaceFunction.forward)")
    b, h, dof = pos.shape
    det = lambda t: t.detach().contiguous()  # noqa: E731
    cost_hip.cspace_state_cost(
        out_cost, out_gp, out_gv, out_ga, out_gj, out_gtau, det(pos), det(vel), det(acc), det(jerk), det(joint_torque), det(state_dt).view(-1),
        det(target_joint_position), det(idxs_target_joint_position).view(-1), p_b, v_b, a_b, j_b, effort_limit, weight,
        activation_distance, squared_l2_regularization_weights, cspace_target_weight, cspace_non_terminal_weight_factor,
        cspace_target_dof_weight, bool(pos.requires_grad), b, h, dof, bool(retime_weights), bool(retime_regularization_weights))
    ctx.save_for_backward(out_gp, out_gv, out_ga, out_gj, out_gtau)
    ctx.use_grad_input = use_grad_input
    ctx.set_materialize_grads(False)
    ctx.mark_non_differentiable(state_dt, target_joint_position, idxs_target_joint_position, p_b, v_b, a_b, j_b, effort_limit, weight,
                                activation_distance, cspace_target_weight, cspace_non_terminal_weight_factor, cspace_target_dof_weight,
                                out_gp, out_gv, out_ga, out_gj, out_gtau)
    return out_cost


def l2_forward(ctx, pos, target, target_idx, weight, terminal_dof_weight, non_terminal_dof_weight, out_cost_dof, out_gp, use_grad_input):
    from curobo_amd.backends import cost as cost_hip

    _count("curobo_hip_cspace_l2_distance (L2DistFunction.forward)")
    b, h, dof = pos.shape
    cost_hip.cspace_l2_distance(out_cost_dof, out_gp, pos.detach().contiguous(), target, target_idx.view(-1), weight, terminal_dof_weight,
                                non_terminal_dof_weight, bool(pos.requires_grad), b, h, dof)
    ctx.save_for_backward(out_gp)
    ctx.use_grad_input = use_grad_input
    return torch.sum(out_cost_dof, dim=-1)


def lm_call(self, state):
    """LevenbergMarquardtStep.__call__ (optim/util/levenberg_marquardt_step.py:96-143): one wavefront per problem, J^T J on
    the matrix cores"""
    from curobo_amd.backends import linalg as linalg_hip

    _count("curobo_hip_levenberg_marquardt_step (LevenbergMarquardtStep.__call__)")
    n, d, r = state.batch_size, self.action_dim, self.n_residuals
    linalg_hip.levenberg_marquardt_step(
        state.joint_position_out.detach().view(n, d), state.pred_reduction.detach().view(n), state.jacobian.detach().view(n, r, d).contiguous(),
        state.jTerror.detach().view(n, d).contiguous(), state.lambda_damping.detach().view(n).contiguous(),
        state.joint_position_in.detach().view(n, d).contiguous())
    return state.joint_position_out, state.pred_reduction


def install():
    """patch every target module that has been imported and not patched yet; returns the names patched now"""
    done = []

    def patch(mod_name, cls_name, fn, attr="forward"):
        mod = sys.modules.get(mod_name)
        if mod is None:
            return
        cls = getattr(mod, cls_name, None)
        if cls is None or getattr(cls, "_curobo_hip_hooked", False):
            return
        setattr(cls, attr, staticmethod(fn))
        cls._curobo_hip_hooked = True
        done.append(f"{mod_name}.{cls_name}.{attr}")

    patch("curobo._src.geom.collision.wp_autograd", "SphereObstacleCollision", sphere_forward)
    patch("curobo._src.geom.collision.wp_autograd", "SweptSphereObstacleCollision", swept_forward)
    patch("curobo._src.cost.wp_tool_pose", "ToolPoseDistance", tool_pose_forward)
    patch("curobo._src.cost.wp_cspace_position", "PositionCSpaceFunction", cspace_position_forward)
    patch("curobo._src.cost.wp_cspace_state", "StateCSpaceFunction", cspace_state_forward)
    patch("curobo._src.cost.wp_torch_cspace_dist", "L2DistFunction", l2_forward)
    mod = sys.modules.get("curobo._src.optim.util.levenberg_marquardt_step")
    if mod is not None and not getattr(mod.LevenbergMarquardtStep, "_curobo_hip_hooked", False):
        mod.LevenbergMarquardtStep.__call__ = lm_call
        mod.LevenbergMarquardtStep._curobo_hip_hooked = True
        done.append("curobo._src.optim.util.levenberg_marquardt_step.LevenbergMarquardtStep.__call__")
    return done
