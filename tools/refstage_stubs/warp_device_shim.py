
# ============================================================================= appended by tools/reference_on_hip.py stage
# Device tensors under the stand-in (GPU box): the emulator computes on host copies, so arrays made from device tensors
# remember their tensor and every launch writes its arrays back.  Only the small set-up kernels of the reference run this
# way (pose algebra while a robot / scene is loaded); the hot-path launches never get here: tools/refstage_stubs/hip_hooks.py
# replaces the bodies of their autograd functions with the calls of INTEGRATION.md (libcurobo_hip.so).
import torch as _torch

_cpu_from_torch = from_torch
_cpu_launch = launch
LAUNCH_LOG = {}  # kernel name -> launches that ran on the host through the stand-in


def from_torch(t, dtype=None, **kw):
    arr = array(t.detach().cpu().numpy(), dtype=dtype)
    if t.is_cuda:
        arr._torch = t
    return arr


def _writeback(x, depth=0):
    if isinstance(x, array):
        t = getattr(x, "_torch", None)
        if t is not None:
            src = _torch.from_numpy(np.ascontiguousarray(x.a))
            t.detach().view(-1).copy_(src.view(-1).to(t.device))
    elif depth < 2 and hasattr(x, "__dict__") and not isinstance(x, (_Vec, Kernel, Function)):
        for v in vars(x).values():
            _writeback(v, depth + 1)


def launch(kernel, dim, inputs=(), outputs=(), device=None, stream=None, **kw):  # noqa: A002
    name = getattr(kernel, "__name__", str(kernel))
    LAUNCH_LOG[name] = LAUNCH_LOG.get(name, 0) + 1
    _cpu_launch(kernel, dim, inputs=inputs, outputs=outputs)
    for a in list(inputs) + list(outputs):
        _writeback(a)


# ---- 3 x 3 matrices for the reference's set-up pose algebra (geom/transform.py: quaternion <-> matrix kernels); row-major,
# stored as a 9-vector so that arrays of them work like arrays of vectors
class mat33(_Vec):
    N = 9

    def __getitem__(self, i):
        if isinstance(i, tuple):
            return self.v[int(i[0]) * 3 + int(i[1])]
        return self.v[int(i)]


mat33f = mat33


def quat_to_matrix(q):  # warp/native/quat.h: q = (x, y, z, w)
    x, y, z, w = (_F(c) for c in q.v)
    two = _F(2.0)
    return mat33(_F(1.0) - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w),
                 two * (x * y + z * w), _F(1.0) - two * (x * x + z * z), two * (y * z - x * w),
                 two * (x * z - y * w), two * (y * z + x * w), _F(1.0) - two * (x * x + y * y))


def quat_from_matrix(m):  # warp/native/quat.h quat_from_matrix
    g = lambda r, c: _F(m.v[r * 3 + c])  # noqa: E731
    tr = g(0, 0) + g(1, 1) + g(2, 2)
    if tr >= 0.0:
        h = _F(np.sqrt(tr + _F(1.0)))
        w = _F(0.5) * h
        h = _F(0.5) / h
        x, y, z = (g(2, 1) - g(1, 2)) * h, (g(0, 2) - g(2, 0)) * h, (g(1, 0) - g(0, 1)) * h
    else:
        i = 0
        if g(1, 1) > g(0, 0):
            i = 1
        if g(2, 2) > g(i, i):
            i = 2
        if i == 0:
            h = _F(np.sqrt((g(0, 0) - (g(1, 1) + g(2, 2))) + _F(1.0)))
            x = _F(0.5) * h
            h = _F(0.5) / h
            y, z, w = (g(0, 1) + g(1, 0)) * h, (g(2, 0) + g(0, 2)) * h, (g(2, 1) - g(1, 2)) * h
        elif i == 1:
            h = _F(np.sqrt((g(1, 1) - (g(2, 2) + g(0, 0))) + _F(1.0)))
            y = _F(0.5) * h
            h = _F(0.5) / h
            z, x, w = (g(1, 2) + g(2, 1)) * h, (g(0, 1) + g(1, 0)) * h, (g(0, 2) - g(2, 0)) * h
        else:
            h = _F(np.sqrt((g(2, 2) - (g(0, 0) + g(1, 1))) + _F(1.0)))
            z = _F(0.5) * h
            h = _F(0.5) / h
            x, y, w = (g(2, 0) + g(0, 2)) * h, (g(1, 2) + g(2, 1)) * h, (g(1, 0) - g(0, 1)) * h
    return normalize(quat(x, y, z, w))
