"""development: what a pure store stream reaches on this GPU (fill / copy), for buffer sizes around the FK outputs"""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize(); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for mb in (14, 28, 57, 114, 456, 1824):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    us_f = t(lambda: x.fill_(1.0)); us_c = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB: fill {us_f:8.2f} us = {mb * 1.048576 / us_f:6.2f} TB/s ; copy {us_c:8.2f} us = {2 * mb * 1.048576 / us_c:6.2f} TB/s (read+write)")
