"""development: duration of the optimiser-side kernel (line search + two-loop + next candidates) vs history / batch"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.optim import LBFGSOpt, LBFGSOptCfg
from curobo_amd.backends import optimization as O

dev = torch.device("cuda:0")
def rollout(x):
    return cost, grad
for B, hist in ((64, 27), (256, 27), (64, 1), (64, 8), (64, 16), (1024, 27)):
    V, N = 84, 4
    cost, grad = torch.rand(B * N, device=dev), torch.randn(B * N, V, device=dev)
    o = LBFGSOpt(LBFGSOptCfg(num_problems=B, history=hist), rollout, 12, 7, (-torch.ones(7, device=dev) * 3, torch.ones(7, device=dev) * 3), dev)
    o.reinitialize(torch.randn(B, 12, 7, device=dev) * 0.1)
    c = o.cfg
    def tail():
        O.launch_lbfgs_iteration_tail(
            o.best_cost, o.best_action, o.best_iteration, o.current_iteration, o.converged, c.convergence_iteration, c.cost_delta_threshold,
            c.cost_relative_threshold, o.exploration_cost, o.exploration_action, o.exploration_gradient, o.exploration_idx.view(-1), o.cost,
            o.action, o.gradient, o.selected_idx.view(-1), o.search_cost, o.x_set, o.search_gradient, o.step_scaled, o._alphas,
            c.line_search_c_1, c.line_search_c_2, False, True, N, V, B, o.step_direction, o.rho, o.y, o.s, o.x_0, o.grad_0, c.epsilon,
            o.history, c.stable_mode, o._step_max, 7, True)
    for _ in range(3): tail()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): tail()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"problems {B:5d} history {hist:2d}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
# floor: 20 dependent launches of a one-element kernel in the same kind of graph
z = torch.zeros(1, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): z.add_(1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"one-element kernel: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
