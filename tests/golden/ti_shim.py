"""Primitive-op stand-in for the ``taichi`` module (fixture generation only, build container only).

Taichi is not installed in the build image and cannot be (no network), so the reference's
``@ti.kernel`` / ``@ti.func`` bodies (render.py:2389-3489) could not be executed in round 1 and the
hot-path oracle stayed unpinned.  This module gives those *unmodified function objects* something
to run on: registered as ``sys.modules["taichi"]`` before ``import render``, it makes the
decorators plain Python wrappers and supplies the ~30 entry points the kernels use with the
semantics Taichi documents for them:

* ``ti.kernel``: converts arguments annotated ``ti.f32`` / ``ti.i32`` to the default float / int
  type, then calls the function; ``ti.func``: calls the function (arguments by value: all values
  here are immutable).  Both count calls and can report them to an observer (instrumentation for
  the fixtures: per-pixel step counts, escape directions, shading inputs -- no kernel logic).
* ``ti.field`` / ``ti.Vector.field``: NumPy-backed; iterating a field yields its indices
  (struct-for); ``to_numpy`` / ``from_numpy`` / ``shape`` / ``[None]`` for 0-d fields.
* ``ti.Vector``: immutable small vector: ``+ - * /`` (element-wise, scalars broadcast), unary
  minus, indexing, ``dot``, ``cross``, ``norm`` (= sqrt of the left-to-right sum of squares),
  ``normalized`` (= ``(1 / norm) * v``, taichi's definition with eps = 0).
* ``ti.cast(x, ti.i32)`` truncates toward zero; integer ``%`` is Python's own (floored), which is
  what Taichi specifies; ``ti.pow`` -- and the ``**`` operator on f32 values, which Taichi maps to it --
  with an integer exponent multiplies (Taichi demotes integer powers to multiplications), otherwise
  it is ``pow``.
* ``ti.sqrt/exp/log/sin/cos/tan/acos/atan2/floor/abs/min/max``, ``ti.math.pi``, ``ti.math.clamp``,
  ``ti.ndrange``, ``ti.init`` (no-op), ``ti.template``, ``ti.f32``, ``ti.i32``, ``ti.cpu``, ``ti.gpu``.

Two arithmetic modes (``set_default_fp``):

* ``"f64"``: what ``ti.init(default_fp=ti.f64)`` would mean: literals and intermediates are Python floats
  (binary64), while everything the reference *declares* f32 stays f32 -- kernel arguments annotated
  ``ti.f32``, ``ti.cast(x, ti.f32)`` and stores into ``dtype=ti.f32`` fields round to binary32.  This is the
  rounding-free value of the reference's statements on the same typed inputs;
* ``"f32"``: every value is a ``numpy.float32`` scalar, so each ``+ - * /`` rounds to binary32 the way
  Taichi's default ``default_fp = f32`` does with IEEE arithmetic (no fast-math re-association);
  sqrt and the transcendentals are computed in binary64 and rounded once to binary32.

What this is NOT: Taichi's code generator.  LLVM fast-math re-association, FMA contraction and the
vendor libm of a real Taichi build are not modelled; the fixtures therefore pin the *statements* of
the reference (formulae, constants, branch structure, index arithmetic, operation order), which is
what a restatement can get wrong.  Nothing in this file knows anything about ray marching.
"""
import itertools
import math as _m
import sys
import types

import numpy as np

# ----------------------------------------------------------------------------- types / mode


class _DType:
    def __init__(self, name, is_float):
        self.name, self.is_float = name, is_float

    def __repr__(self):
        return "ti." + self.name


f32 = _DType("f32", True)
f64 = _DType("f64", True)
i32 = _DType("i32", False)
i64 = _DType("i64", False)
cpu, gpu = "cpu", "gpu"
_TEMPLATE = object()

_MODE = "f64"


def _wrap(name):
    base = getattr(np.float32, name)

    def op(self, other):
        r = base(self, other)
        return _F32(r) if type(r) is np.float32 else r      # NotImplemented (Vector operand) passes through

    op.__name__ = name
    return op


class _F32(np.float32):
    """The f32 scalar of the "f32" mode: numpy.float32 with one difference -- ``x ** n`` with an integer n multiplies
    (``pow`` below), as Taichi's ``**`` does (it is ``ti.pow``, and integer powers are demoted to multiplications),
    where numpy.float32.__pow__ would call powf (render.py:2452, 2574, 2606: ``r_safe ** 3``).  Every arithmetic result
    is again an _F32, so the rule holds wherever a ``**`` is applied."""
    __slots__ = ()

    def __pow__(self, other, mod=None):
        return pow(self, other)

    def __rpow__(self, other, mod=None):
        return pow(other, self)

    def __neg__(self):
        return _F32(np.float32.__neg__(self))

    def __pos__(self):
        return self

    def __abs__(self):
        return _F32(np.float32.__abs__(self))


for _name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__",
              "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__"):
    setattr(_F32, _name, _wrap(_name))


def set_default_fp(mode):
    """"f64": Python floats; "f32": numpy.float32 scalars."""
    global _MODE
    assert mode in ("f64", "f32")
    _MODE = mode


def default_fp():
    return _MODE


def _fp(x):
    """A real number in the default float type."""
    if _MODE == "f32":
        return x if type(x) is _F32 else _F32(x)
    return x if type(x) is float else float(x)


def _as_f32(x):
    """A value declared f32 by the reference (argument annotation, explicit cast): rounded to binary32, then
    carried in the default float type."""
    if _MODE == "f32":
        return x if type(x) is _F32 else _F32(x)
    return float(_F32(x))


def _c(x):
    """Bring a value that may be a Python literal into the default type (ints stay ints)."""
    if isinstance(x, (float, np.floating)):
        return _fp(x)
    if isinstance(x, np.integer):
        return int(x)
    return x


def template():
    return _TEMPLATE


def init(*a, **k):
    return None


# ----------------------------------------------------------------------------- instrumentation

call_counts = {}
_observers = {}
iter_hook = None  # callable(index) invoked before each struct-for iteration of a field


def observe(name, fn):
    """fn(args, result) is called after every call of the @ti.func named ``name`` (None removes it)."""
    if fn is None:
        _observers.pop(name, None)
    else:
        _observers[name] = fn


def func(fn):
    name = fn.__name__
    call_counts.setdefault(name, 0)

    def wrapper(*args):
        call_counts[name] += 1
        out = fn(*args)
        ob = _observers.get(name)
        if ob is not None:
            ob(args, out)
        return out

    wrapper.__name__ = name
    wrapper.__wrapped__ = fn
    return wrapper


def kernel(fn):
    names = fn.__code__.co_varnames[:fn.__code__.co_argcount]
    ann = fn.__annotations__

    def wrapper(*args):
        assert len(args) == len(names), (fn.__name__, len(args), len(names))
        conv = []
        for n, a in zip(names, args):
            t = ann.get(n)
            if isinstance(t, _DType):
                a = (_as_f32(a) if t is f32 else _fp(a)) if t.is_float else int(a)
            conv.append(a)
        return fn(*conv)

    wrapper.__name__ = fn.__name__
    wrapper.__wrapped__ = fn
    return wrapper


# ----------------------------------------------------------------------------- Vector


class Vector:
    __slots__ = ("v",)
    __array_ufunc__ = None  # numpy scalars defer to our reflected operators

    def __init__(self, comps):
        self.v = tuple(_c(x) for x in comps)

    @staticmethod
    def _raw(t):
        o = Vector.__new__(Vector)
        o.v = t
        return o

    def __len__(self):
        return len(self.v)

    def __getitem__(self, i):
        return self.v[i]

    def __iter__(self):
        return iter(self.v)

    def __repr__(self):
        return "Vector(%s)" % (list(self.v),)

    def _zip(self, o):
        if isinstance(o, Vector):
            assert len(o.v) == len(self.v)
            return zip(self.v, o.v)
        o = _c(o)
        return ((a, o) for a in self.v)

    def __add__(self, o):
        return Vector._raw(tuple(a + b for a, b in self._zip(o)))

    __radd__ = __add__

    def __sub__(self, o):
        return Vector._raw(tuple(a - b for a, b in self._zip(o)))

    def __rsub__(self, o):
        return Vector._raw(tuple(b - a for a, b in self._zip(o)))

    def __mul__(self, o):
        return Vector._raw(tuple(a * b for a, b in self._zip(o)))

    def __rmul__(self, o):
        return Vector._raw(tuple(b * a for a, b in self._zip(o)))

    def __truediv__(self, o):
        return Vector._raw(tuple(a / b for a, b in self._zip(o)))

    def __rtruediv__(self, o):
        return Vector._raw(tuple(b / a for a, b in self._zip(o)))

    def __neg__(self):
        return Vector._raw(tuple(-a for a in self.v))

    def dot(self, o):
        it = iter(self._zip(o))
        a, b = next(it)
        s = a * b
        for a, b in it:
            s = s + a * b
        return s

    def norm_sqr(self):
        return self.dot(self)

    def norm(self):
        return sqrt(self.norm_sqr())

    def normalized(self):
        invlen = 1.0 / self.norm()
        return invlen * self

    def cross(self, o):
        a, b = self.v, o.v
        assert len(a) == 3 and len(b) == 3
        return Vector._raw((a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]))


# ----------------------------------------------------------------------------- fields


def _np_dtype(dt):
    if dt.is_float:
        return np.float32 if dt is f32 else np.float64
    return np.int32 if dt is i32 else np.int64


def _indices(shape):
    if len(shape) == 1:
        return range(shape[0])
    return itertools.product(*[range(s) for s in shape])


class _Field:
    def __init__(self, dtype, shape, n=None):
        if isinstance(shape, int):
            shape = (shape,)
        self.dtype, self.shape, self.n = dtype, tuple(int(s) for s in shape), n
        self.arr = np.zeros(self.shape + ((n,) if n else ()), dtype=_np_dtype(dtype))

    def __iter__(self):
        for idx in _indices(self.shape):
            if iter_hook is not None:
                iter_hook(idx)
            yield idx

    def _scalar(self, x):
        return _fp(x) if self.dtype.is_float else int(x)

    def __getitem__(self, idx):
        if idx is None:
            idx = ()
        x = self.arr[idx]
        if self.n:
            assert x.shape == (self.n,), "partial index into a vector field"
            return Vector._raw(tuple(self._scalar(c) for c in x))
        assert x.shape == (), "partial index into a field"
        return self._scalar(x)

    def __setitem__(self, idx, val):
        if idx is None:
            idx = ()
        if isinstance(val, Vector):
            val = val.v
        self.arr[idx] = val

    def from_numpy(self, a):
        a = np.asarray(a)
        assert a.shape == self.arr.shape, (a.shape, self.arr.shape)
        self.arr[...] = a

    def to_numpy(self):
        return self.arr.copy()


def field(dtype, shape=()):
    return _Field(dtype, shape)


def _vector_field(n, dtype, shape=()):
    return _Field(dtype, shape, n)


Vector.field = staticmethod(_vector_field)


def ndrange(*dims):
    if len(dims) == 1:
        return range(int(dims[0]))
    return itertools.product(*[range(int(d)) for d in dims])


# ----------------------------------------------------------------------------- scalar functions


def cast(x, dt):
    if dt.is_float:
        return _as_f32(x) if dt is f32 else _fp(x)
    return int(x)  # truncation toward zero


def _f(x):
    """Argument of a libm call: the value as the default type holds it, widened to binary64."""
    return float(_fp(x))


def sqrt(x):
    return _fp(_m.sqrt(_f(x)))


def exp(x):
    return _fp(_m.exp(_f(x)))


def log(x):
    return _fp(_m.log(_f(x)))


def sin(x):
    return _fp(_m.sin(_f(x)))


def cos(x):
    return _fp(_m.cos(_f(x)))


def tan(x):
    return _fp(_m.tan(_f(x)))


def acos(x):
    return _fp(_m.acos(_f(x)))


def atan2(y, x):
    return _fp(_m.atan2(_f(y), _f(x)))


def floor(x):
    return _fp(_m.floor(_f(x)))


def abs(x):  # noqa: A001
    return -x if x < 0 else x


def pow(a, b):  # noqa: A001
    if isinstance(b, (int, np.integer)) and not isinstance(b, bool):
        b = int(b)
        assert b >= 0
        a = _c(a)
        result, base = None, a
        while b:  # binary exponentiation, as taichi's demotion of integer powers
            if b & 1:
                result = base if result is None else result * base
            b >>= 1
            if b:
                base = base * base
        return _fp(1.0) if result is None else result
    return _fp(_m.pow(_f(a), _f(b)))


def _minmax(a, b, want_max):
    if isinstance(a, Vector) or isinstance(b, Vector):
        v = a if isinstance(a, Vector) else b
        o = b if isinstance(a, Vector) else a
        return Vector._raw(tuple(_minmax(x, y, want_max) for x, y in v._zip(o)))
    a, b = _c(a), _c(b)
    if want_max:
        return a if a >= b else b
    return a if a <= b else b


def max(a, b):  # noqa: A001
    return _minmax(a, b, True)


def min(a, b):  # noqa: A001
    return _minmax(a, b, False)


math = types.ModuleType("taichi.math")
math.pi = _m.pi


def _clamp(x, lo, hi):
    return min(max(x, lo), hi)


math.clamp = _clamp


def install():
    """Register this module as ``taichi`` (and empty ``imageio``) so that ``import render`` works."""
    me = sys.modules[__name__]
    sys.modules["taichi"] = me
    sys.modules["taichi.math"] = math
    for name in ("imageio", "imageio.v3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["imageio"].v3 = sys.modules["imageio.v3"]
    return me
