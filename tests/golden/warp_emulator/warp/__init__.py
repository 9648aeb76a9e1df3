"""A small pure-Python stand-in for the parts of NVIDIA Warp that the reference's collision and cost kernels use.

Purpose: the reference's sphere-obstacle, swept and cost kernels are Warp
kernels, i.e. typed Python functions that Warp compiles for the GPU.  Warp is not installed here and cannot be (no
network), so those kernels never ran next to the oracle.  Their SOURCE is plain Python though: with this module on
``sys.path`` as ``warp``, the reference's own, unmodified kernel functions are imported and executed thread by thread on
the CPU, in fp32, and their outputs become golden vectors (``tests/golden/make_scene_warp_golden.py``).  What is restated
here are only Warp's intrinsics (vector / quaternion / transform algebra, C-style integer division, atomics, function
overloads by struct type): a few lines each, from their published definitions; every branch, activation rule, sweep
loop and accumulation order that is executed is the reference's.

Test infrastructure only: nothing under ``curobo_amd/`` imports this.
"""
import builtins
import math
import types
import typing

import numpy as np

_F = np.float32


# ----------------------------------------------------------------------------- scalars
class _Int(int):
    """C-like integer: ``/`` truncates towards zero, results stay of this class."""

    def _w(self, v):
        return NotImplemented if v is NotImplemented else type(self)(v)

    def __add__(self, o):
        return self._w(int.__add__(self, o)) if isinstance(o, int) else NotImplemented

    __radd__ = __add__

    def __sub__(self, o):
        return self._w(int.__sub__(self, o)) if isinstance(o, int) else NotImplemented

    def __rsub__(self, o):
        return self._w(int.__rsub__(self, o)) if isinstance(o, int) else NotImplemented

    def __mul__(self, o):
        return self._w(int.__mul__(self, o)) if isinstance(o, int) else NotImplemented

    __rmul__ = __mul__

    def __neg__(self):
        return self._w(int.__neg__(self))

    def __truediv__(self, o):
        if not isinstance(o, int):
            return NotImplemented
        q = builtins.abs(int(self)) // builtins.abs(int(o))
        return self._w(q if (int(self) >= 0) == (int(o) >= 0) else -q)

    def __rtruediv__(self, o):
        if not isinstance(o, int):
            return NotImplemented
        return _Int.__truediv__(type(self)(o), self)

    __floordiv__ = __truediv__
    __rfloordiv__ = __rtruediv__

    def __mod__(self, o):  # C remainder
        return self._w(int(math.fmod(int(self), int(o)))) if isinstance(o, int) else NotImplemented

    def __rmod__(self, o):
        return self._w(int(math.fmod(int(o), int(self)))) if isinstance(o, int) else NotImplemented


def _int_type(name):
    def new(cls, v=0):
        if isinstance(v, (np.floating, float)):
            v = math.trunc(float(v))  # C conversion
        return int.__new__(cls, int(v))

    return type(name, (_Int,), {"__new__": new})


int8, int16, int32, int64 = (_int_type(n) for n in ("int8", "int16", "int32", "int64"))
uint8, uint16, uint32, uint64 = (_int_type(n) for n in ("uint8", "uint16", "uint32", "uint64"))
float32, float64, float16 = np.float32, np.float64, np.float16
bool = np.bool_  # noqa: A001  (Warp's name)


def _is_float(x):
    return isinstance(x, (float, np.floating))


# ----------------------------------------------------------------------------- vectors
class _Vec:
    N = 0
    INT = False
    __array_ufunc__ = None  # numpy scalars on the left defer to __rmul__ / __radd__ ... instead of broadcasting over us

    def __init__(self, *a):
        n = self.N
        if len(a) == 0:
            v = [0] * n
        elif len(a) == 1 and isinstance(a[0], _Vec):
            v = list(a[0].v)
        elif len(a) == 1 and not hasattr(a[0], "__len__"):
            v = [a[0]] * n
        elif len(a) == 1:
            v = list(a[0])
        else:
            v = []
            for x in a:  # (vec3, w) style constructors
                v.extend(x.v if isinstance(x, _Vec) else [x])
        assert len(v) == n, (type(self).__name__, a)
        self.v = [int32(x) for x in v] if self.INT else [_F(x) for x in v]

    def __getitem__(self, i):
        return self.v[i]

    def __setitem__(self, i, x):
        self.v[i] = int32(x) if self.INT else _F(x)

    def __len__(self):
        return self.N

    def __iter__(self):
        return iter(self.v)

    def _bin(self, o, f):
        if isinstance(o, _Vec):
            assert o.N == self.N
            return type(self)([f(a, b) for a, b in zip(self.v, o.v)])
        return type(self)([f(a, o) for a in self.v])

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    def __sub__(self, o):
        return self._bin(o, lambda a, b: a - b)

    def __mul__(self, o):
        assert not isinstance(o, _Vec), "Warp has no vec * vec (use cw_mul)"
        return self._bin(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def __truediv__(self, o):
        assert not isinstance(o, _Vec)
        return self._bin(o, lambda a, b: a / b)

    def __neg__(self):
        return type(self)([-a for a in self.v])

    def __pos__(self):
        return type(self)(self.v)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(str(x) for x in self.v)})"

    x = property(lambda s: s.v[0], lambda s, val: s.__setitem__(0, val))
    y = property(lambda s: s.v[1], lambda s, val: s.__setitem__(1, val))
    z = property(lambda s: s.v[2], lambda s, val: s.__setitem__(2, val))
    w = property(lambda s: s.v[3], lambda s, val: s.__setitem__(3, val))


def _vec_type(name, n, integer=False):
    return type(name, (_Vec,), {"N": n, "INT": integer})


vec2, vec3, vec4 = _vec_type("vec2", 2), _vec_type("vec3", 3), _vec_type("vec4", 4)
vec2f, vec3f, vec4f = vec2, vec3, vec4
vec2i, vec3i, vec4i = _vec_type("vec2i", 2, True), _vec_type("vec3i", 3, True), _vec_type("vec4i", 4, True)


class quat(_Vec):  # (x, y, z, w)
    N = 4

    def __mul__(self, o):
        if isinstance(o, quat):  # Hamilton product
            a, b = self.v, o.v
            return quat(a[3] * b[0] + b[3] * a[0] + a[1] * b[2] - b[1] * a[2],
                        a[3] * b[1] + b[3] * a[1] + a[2] * b[0] - b[2] * a[0],
                        a[3] * b[2] + b[3] * a[2] + a[0] * b[1] - b[0] * a[1],
                        a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2])
        return _Vec.__mul__(self, o)


quatf = quat
quaternion = quat


class transform:
    """(translation p, rotation q)."""

    __array_ufunc__ = None

    def __init__(self, p=None, q=None):
        self.p = vec3() if p is None else vec3(p)
        self.q = quat(0.0, 0.0, 0.0, 1.0) if q is None else quat(q)

    def __mul__(self, o):
        return transform_multiply(self, o)

    def __repr__(self):
        return f"transform({self.p}, {self.q})"


transformf = transform


def dot(a, b):
    s = _F(0.0)
    for x, y in zip(a.v, b.v):
        s = s + x * y
    return s


def cross(a, b):
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def length(a):
    return _F(np.sqrt(dot(a, a)))


def length_sq(a):
    return dot(a, a)


def normalize(a):
    n = length(a)
    return a / n if n > 0 else type(a)()


def mul(a, b):
    return a * b


_generic_vecs = {}


def vector(*vals, length=None, dtype=None):
    """``wp.vector(a, b, ...)``: a float vector of as many components as arguments"""
    n = len(vals) if vals else int(length)
    cls = _generic_vecs.get(n) or _generic_vecs.setdefault(n, {2: vec2, 3: vec3, 4: vec4}.get(n) or _vec_type(f"vec{n}", n))
    return cls(*vals) if vals else cls()


def cw_mul(a, b):
    return type(a)([x * y for x, y in zip(a.v, b.v)])


def cw_div(a, b):
    return type(a)([x / y for x, y in zip(a.v, b.v)])


class mat33(_Vec):  # row major; m[i, j] or m[i][j]
    N = 9

    def __init__(self, *a):
        if len(a) == 3 and all(isinstance(x, _Vec) and x.N == 3 for x in a):  # from three column vectors (Warp's convention)
            c0, c1, c2 = a
            a = ([c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2]],)
        _Vec.__init__(self, *a)

    def __getitem__(self, i):
        if isinstance(i, tuple):
            return self.v[int(i[0]) * 3 + int(i[1])]
        return vec3(self.v[3 * int(i):3 * int(i) + 3])

    def __setitem__(self, i, x):
        assert isinstance(i, tuple)
        self.v[int(i[0]) * 3 + int(i[1])] = _F(x)

    def __mul__(self, o):
        if isinstance(o, mat33):
            return mat33([sum((self[i, k] * o[k, j] for k in range(1, 3)), self[i, 0] * o[0, j]) for i in range(3) for j in range(3)])
        if isinstance(o, _Vec) and o.N == 3:
            return vec3([self[i, 0] * o[0] + self[i, 1] * o[1] + self[i, 2] * o[2] for i in range(3)])
        return _Vec.__mul__(self, o)


mat33f = mat33


def quat_to_matrix(q):  # warp/native/quat.h: the columns are the rotated unit vectors
    return mat33(quat_rotate(q, vec3(1.0, 0.0, 0.0)), quat_rotate(q, vec3(0.0, 1.0, 0.0)), quat_rotate(q, vec3(0.0, 0.0, 1.0)))


def quat_from_matrix(m):  # warp/native/quat.h: trace form, else the largest diagonal element; normalised
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.v
    tr = m00 + m11 + m22
    half, one = _F(0.5), _F(1.0)
    if tr >= _F(0.0):
        h = _F(np.sqrt(tr + one))
        w = half * h
        h = half / h
        x, y, z = (m21 - m12) * h, (m02 - m20) * h, (m10 - m01) * h
    else:
        d = 0
        if m11 > m00:
            d = 1
        if m22 > (m00, m11)[d]:
            d = 2
        if d == 0:
            h = _F(np.sqrt((m00 - (m11 + m22)) + one))
            x = half * h
            h = half / h
            y, z, w = (m01 + m10) * h, (m20 + m02) * h, (m21 - m12) * h
        elif d == 1:
            h = _F(np.sqrt((m11 - (m22 + m00)) + one))
            y = half * h
            h = half / h
            z, x, w = (m12 + m21) * h, (m01 + m10) * h, (m02 - m20) * h
        else:
            h = _F(np.sqrt((m22 - (m00 + m11)) + one))
            z = half * h
            h = half / h
            x, y, w = (m20 + m02) * h, (m12 + m21) * h, (m10 - m01) * h
    return normalize(quat(x, y, z, w))


def quat_identity():
    return quat(0.0, 0.0, 0.0, 1.0)


def quat_inverse(q):
    return quat(-q[0], -q[1], -q[2], q[3])


def quat_rotate(q, v):  # warp/native/quat.h
    qv = vec3(q[0], q[1], q[2])
    return v * (_F(2.0) * q[3] * q[3] - _F(1.0)) + cross(qv, v) * q[3] * _F(2.0) + qv * dot(qv, v) * _F(2.0)


def quat_rotate_inv(q, v):
    qv = vec3(q[0], q[1], q[2])
    return v * (_F(2.0) * q[3] * q[3] - _F(1.0)) - cross(qv, v) * q[3] * _F(2.0) + qv * dot(qv, v) * _F(2.0)


def transform_identity():
    return transform()


def transform_get_translation(t):
    return vec3(t.p)


def transform_get_rotation(t):
    return quat(t.q)


def transform_point(t, p):
    return quat_rotate(t.q, p) + t.p


def transform_vector(t, v):
    return quat_rotate(t.q, v)


def transform_inverse(t):
    qi = quat_inverse(t.q)
    return transform(-quat_rotate(qi, t.p), qi)


def transform_multiply(a, b):
    return transform(quat_rotate(a.q, b.p) + a.p, a.q * b.q)


# ----------------------------------------------------------------------------- scalar math
def _f1(fn):
    return lambda x: _F(fn(_F(x)))


sqrt, sin, cos, tan, exp, log, floor, ceil = (_f1(f) for f in (np.sqrt, np.sin, np.cos, np.tan, np.exp, np.log, np.floor, np.ceil))
acos, asin, atan, tanh = (_f1(f) for f in (np.arccos, np.arcsin, np.arctan, np.tanh))


def atan2(y, x):
    return _F(np.arctan2(_F(y), _F(x)))


def pow(x, y):  # noqa: A001
    return _F(np.power(_F(x), _F(y)))


def _pick(a, b, take_a):
    if isinstance(a, _Vec):
        return type(a)([_pick(x, y, take_a) for x, y in zip(a.v, b.v if isinstance(b, _Vec) else [b] * a.N)])
    r = a if take_a(a, b) else b
    if _is_float(a) or _is_float(b):
        return _F(r)
    return r if isinstance(r, _Int) else int32(r)


def max(a, b):  # noqa: A001
    return _pick(a, b, lambda x, y: x > y)


def min(a, b):  # noqa: A001
    return _pick(a, b, lambda x, y: x < y)


def abs(a):  # noqa: A001
    if isinstance(a, _Vec):
        return type(a)([abs(x) for x in a.v])
    return _F(np.abs(a)) if _is_float(a) else type(a)(-a if a < 0 else a)


def sign(a):
    return _F(-1.0) if a < 0 else _F(1.0)


def clamp(x, lo, hi):
    return min(max(x, lo), hi)


def select(cond, a, b):  # select(cond, value_if_false, value_if_true)
    return b if cond else a


def where(cond, a, b):
    return a if cond else b


def isnan(x):
    return np.isnan(x)


def isfinite(x):
    return np.isfinite(x)


# ----------------------------------------------------------------------------- arrays
class _ArrayAnnotation:
    def __init__(self, dtype=None, ndim=1):
        self.dtype, self.ndim = dtype, ndim


class array:
    """``wp.array(dtype=...)`` in an annotation; ``wp.array(numpy, dtype=...)`` / ``from_numpy`` for data.
    Element reads return Warp-typed scalars / vectors, writes go through to the numpy storage."""

    def __new__(cls, data=None, dtype=None, ndim=1, **kw):
        if data is None:
            return _ArrayAnnotation(dtype, ndim)
        return object.__new__(cls)

    def __init__(self, data=None, dtype=None, ndim=1, **kw):
        a = np.asarray(data)
        self.dtype = dtype if dtype is not None else {np.dtype(np.float32): float32, np.dtype(np.int32): int32,
                                                       np.dtype(np.uint8): uint8, np.dtype(np.float16): float16,
                                                       np.dtype(np.int16): int16, np.dtype(np.int64): int64}[a.dtype]
        self.vec = isinstance(self.dtype, type) and issubclass(self.dtype, _Vec)
        if self.vec and (a.ndim == 0 or a.shape[-1] != self.dtype.N):
            a = a.reshape(-1, self.dtype.N)
        self.a = a
        self.shape = a.shape[:-1] if self.vec else a.shape
        self.ndim = len(self.shape)

    def _wrap(self, x):
        d = self.dtype
        if self.vec:
            return d(list(x))
        if d in (np.float32, np.float64, np.float16):
            return d(x)
        return d(int(x))

    def __getitem__(self, idx):
        x = self.a[idx]
        if isinstance(x, np.ndarray) and x.ndim > (1 if self.vec else 0):  # a row / slab of a 2-d / 3-d array
            return array(x, dtype=self.dtype)
        return self._wrap(x)

    def __setitem__(self, idx, val):
        if self.vec:
            self.a[idx] = np.asarray([float(x) for x in val.v], dtype=self.a.dtype)
        else:
            self.a[idx] = val

    def numpy(self):
        return self.a


def array1d(dtype=None, **kw):
    return _ArrayAnnotation(dtype, 1)


def array2d(dtype=None, **kw):
    return _ArrayAnnotation(dtype, 2)


def array3d(dtype=None, **kw):
    return _ArrayAnnotation(dtype, 3)


def array4d(dtype=None, **kw):
    return _ArrayAnnotation(dtype, 4)


def from_numpy(a, dtype=None, **kw):
    return array(a, dtype=dtype)


def from_torch(t, dtype=None, **kw):
    return array(t.detach().cpu().numpy(), dtype=dtype)


def _slot(arr, idx):
    return arr.a, idx


def atomic_add(arr, *args):
    *idx, val = args
    idx = tuple(idx) if len(idx) > 1 else idx[0]
    old = arr[idx]
    arr[idx] = old + val
    return old


def atomic_sub(arr, *args):
    *idx, val = args
    idx = tuple(idx) if len(idx) > 1 else idx[0]
    old = arr[idx]
    arr[idx] = old - val
    return old


def atomic_max(arr, *args):
    *idx, val = args
    idx = tuple(idx) if len(idx) > 1 else idx[0]
    old = arr[idx]
    arr[idx] = max(old, val)
    return old


def atomic_min(arr, *args):
    *idx, val = args
    idx = tuple(idx) if len(idx) > 1 else idx[0]
    old = arr[idx]
    arr[idx] = min(old, val)
    return old


# ----------------------------------------------------------------------------- tiles (the block-cooperative API)
class Tile:
    """a block-wide register tile: here just an fp32 numpy array"""

    __array_ufunc__ = None

    def __init__(self, a):
        self.a = np.array(a, dtype=np.float32)

    shape = property(lambda s: s.a.shape)

    def __getitem__(self, i):
        x = self.a[i]
        return Tile(x) if isinstance(x, np.ndarray) and x.ndim else _F(x)

    def __setitem__(self, i, v):
        self.a[i] = v.a if isinstance(v, Tile) else v


def _shape(shape):
    return tuple(int(x) for x in shape) if hasattr(shape, "__len__") else (int(shape),)


def tile_load(arr, shape=None, offset=None, **kw):
    a = arr.a if isinstance(arr, array) else np.asarray(arr)
    shp = _shape(shape)
    if offset is not None:
        a = a[tuple(slice(int(o), int(o) + n) for o, n in zip(_shape(offset), shp))]
    return Tile(np.asarray(a, np.float32).reshape(shp))


def tile_store(arr, t, offset=None, **kw):
    dst = arr.a if isinstance(arr, array) else arr
    if offset is not None:
        dst = dst[tuple(slice(int(o), int(o) + n) for o, n in zip(_shape(offset), t.a.shape))]
    dst[...] = t.a.reshape(dst.shape)


def tile_zeros(shape=None, dtype=None, **kw):
    return Tile(np.zeros(_shape(shape), np.float32))


def tile_ones(shape=None, dtype=None, **kw):
    return Tile(np.ones(_shape(shape), np.float32))


def tile_transpose(t):
    return Tile(t.a.T.copy())


def tile_matmul(a, b, out=None):
    """out += a b, every product and sum rounded to fp32 in index order"""
    A, B = a.a, b.a
    C = np.zeros((A.shape[0], B.shape[1]), np.float32) if out is None else out.a
    for i in range(A.shape[0]):
        for j in range(B.shape[1]):
            acc = _F(C[i, j])
            for k in range(A.shape[1]):
                acc = _F(acc + _F(A[i, k] * B[k, j]))
            C[i, j] = acc
    return Tile(C) if out is None else out


def tile_diag_add(a, d):
    r = a.a.copy()
    r[np.arange(r.shape[0]), np.arange(r.shape[0])] += d.a
    return Tile(r)


def tile_map(op, *tiles):
    out = np.empty_like(tiles[0].a)
    flat = [t.a.reshape(-1) for t in tiles]
    o = out.reshape(-1)
    for i in range(o.size):
        o[i] = op(*[_F(f[i]) for f in flat])
    return Tile(out)


def tile_sum(t):
    acc = _F(0.0)
    for x in t.a.reshape(-1):
        acc = _F(acc + x)
    return Tile(np.array([acc], np.float32))


def tile_cholesky(A):
    """lower factor, row by row (Cholesky-Banachiewicz), fp32"""
    a = A.a
    n = a.shape[0]
    L = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in range(i + 1):
            acc = _F(a[i, j])
            for k in range(j):
                acc = _F(acc - _F(L[i, k] * L[j, k]))
            L[i, j] = _F(np.sqrt(acc)) if i == j else _F(acc / L[j, j])
    return Tile(L)


def tile_cholesky_solve(L, y):
    """solve L L^T x = y"""
    l, b = L.a, y.a.reshape(-1)
    n = l.shape[0]
    z = np.zeros(n, np.float32)
    for i in range(n):
        acc = _F(b[i])
        for k in range(i):
            acc = _F(acc - _F(l[i, k] * z[k]))
        z[i] = _F(acc / l[i, i])
    x = np.zeros(n, np.float32)
    for i in range(n - 1, -1, -1):
        acc = _F(z[i])
        for k in range(i + 1, n):
            acc = _F(acc - _F(l[k, i] * x[k]))
        x[i] = _F(acc / l[i, i])
    return Tile(x.reshape(y.a.shape))


def neg(a):
    return -a


def add(a, b):
    return a + b


def sub(a, b):
    return a - b


def launch_tiled(kernel, dim, inputs=(), outputs=(), block_dim=None, **kw):
    return launch(kernel, dim, inputs=inputs, outputs=outputs)


# ----------------------------------------------------------------------------- functions, kernels, structs
_tid = [0]


def tid():
    t = _tid[0]
    return tuple(int32(x) for x in t) if isinstance(t, tuple) else int32(t)


class Function:
    """``@wp.func``: a plain call, or -- when several functions of one name were registered -- the overload whose first
    parameter is annotated with the class of the first argument (Warp resolves overloads by argument types; the
    reference overloads its obstacle accessors by obstacle struct)."""

    def __init__(self, name):
        self.name, self.overloads = name, []

    def add(self, fn):
        if isinstance(fn, Function):
            for f in fn.overloads:
                self.add(f)
        elif fn not in self.overloads:
            self.overloads.append(fn)
        return self

    @staticmethod
    def _first_annotation(fn):
        ann = getattr(fn, "__annotations__", {})
        code = fn.__code__
        first = code.co_varnames[0] if code.co_argcount else None
        a = ann.get(first)
        return a if not isinstance(a, str) else fn.__globals__.get(a, a)

    def __call__(self, *a, **k):
        if len(self.overloads) == 1:
            return self.overloads[0](*a, **k)
        for fn in self.overloads:
            want = self._first_annotation(fn)
            if isinstance(want, type) and isinstance(a[0], want):
                return fn(*a, **k)
            if isinstance(want, str) and type(a[0]).__name__ == want:
                return fn(*a, **k)
        raise TypeError(f"no overload of {self.name} for {type(a[0]).__name__}")


_functions = {}


def func(f=None, *, name=None, module=None, **kw):
    if f is None:
        return lambda g: func(g, name=name, module=module, **kw)
    base = f.name if isinstance(f, Function) else f.__name__
    if module == "unique":  # (Warp: a module of its own per function, never an overload of an earlier one)
        return Function(name or base).add(f)
    key = (module or getattr(f, "__module__", None), name or base)
    reg = _functions.get(key)
    if reg is None:
        reg = _functions[key] = Function(name or base)
    return reg.add(f)


class Kernel:
    def __init__(self, fn):
        self.fn, self.__name__ = fn, fn.__name__

    def __call__(self, *a, **k):
        return self.fn(*a, **k)


def kernel(f=None, **kw):
    if f is None:
        return lambda g: Kernel(g)
    return Kernel(f)


def launch(kernel, dim, inputs=(), outputs=(), device=None, stream=None, **kw):  # noqa: A002
    """Every thread in index order, one after the other (a legal schedule of the GPU launch; outputs that the reference
    accumulates with float atomics therefore come out in thread-index order)."""
    args = [int32(a) if type(a) is int else (_F(a) if type(a) is float else a) for a in list(inputs) + list(outputs)]
    fn = kernel.fn if isinstance(kernel, Kernel) else kernel
    if isinstance(dim, (tuple, list)) and len(dim) > 1:
        import itertools

        for t in itertools.product(*[range(int(d)) for d in dim]):
            _tid[0] = tuple(t)
            fn(*args)
    else:
        n = int(dim[0]) if isinstance(dim, (tuple, list)) else int(dim)
        for t in range(n):
            _tid[0] = t
            fn(*args)


def struct(cls):
    """``@wp.struct``: instances take their fields as keyword arguments or by assignment."""
    ann = dict(getattr(cls, "__annotations__", {}))

    def __init__(self, **kw):
        for k in ann:
            setattr(self, k, None)
        for k, v in kw.items():
            setattr(self, k, v)

    cls.__init__ = __init__
    return cls


def constant(x):
    return x


def static(x):
    return x


def overload(f, *a, **k):
    return f


def init(*a, **k):
    return None


def synchronize(*a, **k):
    return None


class _Config(types.SimpleNamespace):
    pass


config = _Config(quiet=True, mode="release", verify_cuda=False, enable_backward=False, kernel_cache_dir=None,
                 version="1.9.0")  # (the reference only compares it against minimum versions)


class _Anything:
    """Whatever else a module touches at import time (types in annotations, device handles, ...)."""

    def __init__(self, name="warp"):
        self.__dict__["_name"] = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(f"{self._name}.{k}")

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], _Anything):
            return a[0]  # used as a decorator
        return _Anything(f"{self._name}()")

    def __getitem__(self, k):
        return self

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)

    def __repr__(self):
        return f"<warp emulator placeholder {self._name}>"


def __getattr__(name):  # module-level: anything this emulator does not model
    if name.startswith("__") and name.endswith("__"):
        raise AttributeError(name)
    return _Anything(f"warp.{name}")


Any = typing.Any
