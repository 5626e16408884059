"""ti_shim/taichi -- a scalar interpreter for the subset of the Taichi DSL that the reference's hot-path kernels
use (modules/{intersection,ray_march,hash_encoder,hash_encoder_half,spherical_harmonics,volume_train,
volume_render_test,utils,distortion}.py).

TEST INFRASTRUCTURE ONLY (lives under oracle/).  The real `taichi` package is neither installed nor installable
in this environment, so the reference's kernels cannot be launched.  This shim lets oracle/gen_golden.py IMPORT
THE REFERENCE'S OWN SOURCE FILES from /root/reference and execute every `@ti.kernel` body as plain Python with
Taichi's typing rules emulated by numpy scalars:
  * default_fp = f32, default_ip = i32: values read from f32 tensors are np.float32 and stay np.float32 through
    arithmetic with Python literals (numpy 2 "weak scalar" promotion == Taichi's literal typing); Python/numpy
    float64 constants captured by a kernel are demoted to f32 exactly like Taichi bakes captured floats;
  * u32 arithmetic wraps; casts truncate; ti.bit_cast reinterprets;
  * every parallel-for runs serially in index order, so atomics are deterministic (ray order);
  * separate multiply and add (no FMA contraction) -- Taichi's fast_math may contract; the restatement and the
    HIP kernels are defined against the strict evaluation.
It is NOT the Taichi runtime, and nothing here is optimised -- golden cases are a few hundred rays/points.
Round 4: `kernel.grad(...)` is emulated by a reverse-mode tape over the same forward source (taichi/_autodiff.py): tensor
arguments that require grad become taped arrays, output seeds are read from / input adjoints accumulated into their `.grad`,
as Taichi's torch interop does (oracle/gen_golden_autodiff.py makes the backward fixtures with it).
"""
import ast
import inspect
import itertools
import textwrap
import types as _pytypes

import numpy as np

from ._autodiff import AD as _AD, ADArray as _ADArray, Tape as _Tape, ad_max as _ad_max, ad_min as _ad_min

np.seterr(all="ignore")

# ----------------------------------------------------------------------------------------------- dtypes
f32 = np.float32
f16 = np.float16
f64 = np.float64
i32 = np.int32
i64 = np.int64
u8 = np.uint8
u32 = np.uint32
u64 = np.uint64
int32 = np.int32
uint32 = np.uint32
uint8 = np.uint8
float32 = np.float32

cuda = "cuda"
cpu = "cpu"
vulkan = "vulkan"


def init(*args, **kwargs):
    return None


def reset():
    return None


def sync():
    return None


# ----------------------------------------------------------------------------------------------- vectors
class Vector:
    """Small fixed-size vector with element dtype (ti.Vector / vec3 / uvec3 / ti.types.vector)."""
    __array_priority__ = 1000

    def __init__(self, data, dtype=None):
        if isinstance(data, Vector):
            data = data.a
        arr = np.array(data)
        if dtype is None:
            if arr.dtype == np.float64:
                dtype = np.float32
            elif arr.dtype == np.int64:
                dtype = np.int32
            else:
                dtype = arr.dtype
        self.a = arr.astype(dtype)

    # element access
    def __getitem__(self, i):
        return self.a[i]

    def __setitem__(self, i, v):
        if isinstance(v, _AD) and self.a.dtype != object:
            self.a = self.a.astype(object)            # a taped scalar (kernel.grad emulation): element-wise Python arithmetic
        self.a[i] = v

    def __len__(self):
        return len(self.a)

    def _wrap(self, r):
        if isinstance(r, np.ndarray) and r.dtype == np.float64:
            r = r.astype(np.float32)
        return Vector(r, r.dtype)

    @staticmethod
    def _v(o):
        if isinstance(o, Vector):
            return o.a
        if isinstance(o, (np.float64,)):
            return np.float32(o)
        return o

    def __add__(self, o): return self._wrap(self.a + self._v(o))
    def __radd__(self, o): return self._wrap(self._v(o) + self.a)
    def __sub__(self, o): return self._wrap(self.a - self._v(o))
    def __rsub__(self, o): return self._wrap(self._v(o) - self.a)
    def __mul__(self, o): return self._wrap(self.a * self._v(o))
    def __rmul__(self, o): return self._wrap(self._v(o) * self.a)
    def __truediv__(self, o): return self._wrap(self.a / self._v(o))
    def __rtruediv__(self, o): return self._wrap(self._v(o) / self.a)
    def __neg__(self): return self._wrap(-self.a)
    def __lshift__(self, o): return self._wrap(self.a << self._v(o))
    def __or__(self, o): return self._wrap(self.a | self._v(o))
    def __and__(self, o): return self._wrap(self.a & self._v(o))

    # Augmented assignment.  Taichi lowers `x op= v` to an AtomicOpStmt whose type check casts `v` to x's element type BEFORE the
    # operation (its "Atomic add (f32 to f16) may lose precision" warning); the operation then runs in that type.  So an f32
    # product accumulated into an f16 vector (hash_encoder_half.py:212) is rounded to f16 first and the add rounds once more --
    # not one rounding of an f32 sum (round 2's shim did that: 1 ulp off on rows with two contributions, -0 instead of +0 for
    # products that underflow f16).  Same-type operands (every other kernel) are unaffected.
    def _rhs(self, o):
        o = self._v(o)
        return np.asarray(o).astype(self.a.dtype) if isinstance(o, (np.ndarray, np.generic)) else o

    def __iadd__(self, o):
        self.a = (self.a + self._rhs(o)).astype(self.a.dtype); return self

    def __isub__(self, o):
        self.a = (self.a - self._rhs(o)).astype(self.a.dtype); return self

    def __imul__(self, o):
        self.a = (self.a * self._rhs(o)).astype(self.a.dtype); return self

    def max(self):
        r = self.a[0]
        for v in self.a[1:]:
            r = max(r, v)
        return r

    def min(self):
        r = self.a[0]
        for v in self.a[1:]:
            r = min(r, v)
        return r

    def any(self):
        return bool(np.any(self.a != 0))

    def __repr__(self):
        return "Vector(%r)" % (self.a,)


class _VecType:
    def __init__(self, n, dtype):
        self.n, self.dtype = n, dtype

    def __call__(self, *args):
        if len(args) == 1 and not isinstance(args[0], (list, tuple, Vector, np.ndarray)):
            return Vector([args[0]] * self.n, self.dtype)
        if len(args) == 1:
            return Vector(args[0], self.dtype)
        return Vector(list(args), self.dtype)


class _VecArray:
    """ndarray whose elements are vectors (ti.types.ndarray(dtype=vec3, ndim=1) etc.)."""

    def __init__(self, arr):
        self.arr = arr
        self.shape = arr.shape[:-1]

    def __getitem__(self, i):
        return Vector(np.array(self.arr[i]), self.arr.dtype)

    def __setitem__(self, i, v):
        self.arr[i] = v.a if isinstance(v, Vector) else v


# ----------------------------------------------------------------------------------------------- ti.types / ti.math
class _NdAnn:
    def __init__(self, dtype=None, ndim=None):
        self.dtype, self.ndim = dtype, ndim

    @property
    def is_vec(self):
        return isinstance(self.dtype, _VecType)


class _Template:
    pass


def template():
    return _Template()


class _Types:
    @staticmethod
    def ndarray(dtype=None, ndim=None, **kw):
        return _NdAnn(dtype, ndim)

    @staticmethod
    def vector(n, dtype):
        return _VecType(n, dtype)


types = _Types()


def _to_f32(x):
    if isinstance(x, Vector):
        return x
    if isinstance(x, (float, np.floating)):
        return np.float32(x)
    return x


def _is_float(x):
    return isinstance(x, (float, np.floating)) or (isinstance(x, Vector) and x.a.dtype.kind == "f")


def _minmax(fn, a, b):
    if isinstance(a, Vector) or isinstance(b, Vector):
        av = a.a if isinstance(a, Vector) else a
        bv = b.a if isinstance(b, Vector) else b
        r = fn(av, bv)
        return Vector(r, np.float32 if r.dtype.kind == "f" else r.dtype)
    if _is_float(a) or _is_float(b):
        return np.float32(fn(np.float32(a), np.float32(b)))
    return fn(a, b)


def min(a, b):  # noqa: A001
    if isinstance(a, _AD) or isinstance(b, _AD):
        return _ad_min(a, b)
    return _minmax(np.fmin if (_is_float(a) or _is_float(b)) else np.minimum, a, b)


def max(a, b):  # noqa: A001
    if isinstance(a, _AD) or isinstance(b, _AD):
        return _ad_max(a, b)
    return _minmax(np.fmax if (_is_float(a) or _is_float(b)) else np.maximum, a, b)


def abs(x):  # noqa: A001
    return Vector(np.abs(x.a), x.a.dtype) if isinstance(x, Vector) else np.abs(x)


def _unary(fn, x):
    if isinstance(x, Vector):
        return Vector(fn(x.a.astype(np.float32)), np.float32)
    return np.float32(fn(np.float32(x)))


def exp(x): return x.exp() if isinstance(x, _AD) else _unary(np.exp, x)
def log(x): return _unary(np.log, x)
def sqrt(x): return _unary(np.sqrt, x)
def floor(x): return _unary(np.floor, x)
def ceil(x): return _unary(np.ceil, x)


def pow(a, b):  # noqa: A001
    return np.float32(np.power(np.float32(a), np.float32(b)))


def cast(x, dtype):
    if isinstance(x, Vector):
        if np.dtype(dtype).kind in "ui" and x.a.dtype.kind == "f":
            return Vector(np.trunc(x.a).astype(np.int64).astype(dtype), dtype)
        return Vector(x.a.astype(dtype), dtype)
    if np.dtype(dtype).kind in "ui" and isinstance(x, (float, np.floating)):
        return np.dtype(dtype).type(int(x))
    if np.dtype(dtype).kind in "ui":
        return np.dtype(dtype).type(int(x) & ((1 << (8 * np.dtype(dtype).itemsize)) - 1)) if np.dtype(dtype).kind == "u" \
            else np.dtype(dtype).type(int(x))
    return np.dtype(dtype).type(x)


def bit_cast(x, dtype):
    return np.array([x]).view(dtype)[0] if not isinstance(x, Vector) else Vector(x.a.view(dtype), dtype)


def random(dtype=float):
    return np.float32(np.random.random())


def static(x):
    return x


def ndrange(*args):
    if len(args) == 1:
        return range(int(args[0]))
    return itertools.product(*[range(int(a)) for a in args])


def grouped(x):
    return itertools.product(*[range(s) for s in x.shape])


def loop_config(**kw):
    return None


def _atomic_add(arr, idx, val):
    old = arr[idx]
    arr[idx] = old + val
    return old


def atomic_add(*a):
    raise RuntimeError("ti.atomic_add must be rewritten by the kernel transformer")


def func(fn):
    return fn


class _Math(_pytypes.ModuleType):
    pass


math = _Math("taichi.math")
math.vec2 = _VecType(2, np.float32)
math.vec3 = _VecType(3, np.float32)
math.uvec3 = _VecType(3, np.uint32)
math.ivec3 = _VecType(3, np.int32)        # (modules/triplane.py imports it; the tri-plane kernels themselves are never run)
math.pow = pow
math.min = min
math.max = max


def _clamp(x, xmin, xmax):
    return min(xmax, max(xmin, x))


def _sign(x):
    if isinstance(x, Vector):
        return Vector(np.sign(x.a), x.a.dtype)
    return np.float32(np.sign(x))


math.clamp = _clamp
math.sign = _sign
math.exp = exp
math.floor = floor


def Vector_ctor(data, dt=None):
    return Vector(data, dt)


# ----------------------------------------------------------------------------------------------- kernels
class _Rewrite(ast.NodeTransformer):
    """ti.atomic_add(x[i], v) -> _ti_atomic_add(x, i, v);  `for i in <ndarray param>` -> range(shape[0])."""

    def __init__(self, nd_params):
        self.nd_params = nd_params

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if isinstance(f, ast.Attribute) and f.attr == "atomic_add" and isinstance(node.args[0], ast.Subscript):
            sub = node.args[0]
            return ast.copy_location(ast.Call(func=ast.Name(id="_ti_atomic_add", ctx=ast.Load()),
                                              args=[sub.value, sub.slice, node.args[1]], keywords=[]), node)
        return node

    def visit_For(self, node):
        self.generic_visit(node)
        if isinstance(node.iter, ast.Name) and node.iter.id in self.nd_params:
            node.iter = ast.Call(func=ast.Name(id="range", ctx=ast.Load()),
                                 args=[ast.Subscript(value=ast.Attribute(value=ast.Name(id=node.iter.id, ctx=ast.Load()),
                                                                         attr="shape", ctx=ast.Load()),
                                                     slice=ast.Constant(0), ctx=ast.Load())], keywords=[])
        return node


def _demote(v):
    if isinstance(v, np.float64):
        return np.float32(v)
    return v


class _Kernel:
    def __init__(self, fn):
        self.fn = fn
        self.__name__ = fn.__name__
        self._compiled = None
        self.sig = inspect.signature(fn)

    def _build(self):
        src = textwrap.dedent(inspect.getsource(self.fn))
        tree = ast.parse(src)
        fdef = tree.body[0]
        fdef.decorator_list = []
        nd = {n for n, p in self.sig.parameters.items() if isinstance(p.annotation, _NdAnn)}
        for a in fdef.args.args:
            a.annotation = None
        fdef.returns = None
        tree = ast.fix_missing_locations(_Rewrite(nd).visit(tree))
        ns = {k: _demote(v) for k, v in self.fn.__globals__.items()}
        if self.fn.__closure__:
            for name, cell in zip(self.fn.__code__.co_freevars, self.fn.__closure__):
                try:
                    ns[name] = _demote(cell.cell_contents)
                except ValueError:
                    pass
        ns["_ti_atomic_add"] = _atomic_add
        exec(compile(tree, inspect.getsourcefile(self.fn) or "<ti_shim>", "exec"), ns)
        self._compiled = ns[self.fn.__name__]

    def __call__(self, *args, **kwargs):
        if self._compiled is None:
            self._build()
        bound = self.sig.bind(*args, **kwargs)
        conv = []
        for name, val in bound.arguments.items():
            ann = self.sig.parameters[name].annotation
            if isinstance(ann, _NdAnn):
                arr = val.detach().numpy() if hasattr(val, "detach") else np.asarray(val)
                conv.append(_VecArray(arr) if ann.is_vec else arr)
            elif ann in (float, np.float32):
                conv.append(np.float32(val))
            elif ann in (int, np.int32):
                conv.append(int(val))
            else:
                conv.append(val)
        return self._compiled(*conv)

    def grad(self, *args, **kwargs):
        """`kernel.grad(...)`: the forward source once more under a reverse-mode tape (taichi/_autodiff.py).  Tensor arguments
        with requires_grad are taped; seeds come from / adjoints go to their `.grad`, like Taichi's torch interop."""
        if self._compiled is None:
            self._build()
        bound = self.sig.bind(*args, **kwargs)
        tape = _Tape()
        conv, taped = [], []
        for name, val in bound.arguments.items():
            ann = self.sig.parameters[name].annotation
            if isinstance(ann, _NdAnn):
                if getattr(val, "requires_grad", False) and val.dtype.is_floating_point:
                    if ann.is_vec:
                        raise NotImplementedError("taped vector-typed ndarrays")
                    w = _ADArray(tape, val)
                    conv.append(w); taped.append(w)
                else:
                    arr = val.detach().numpy() if hasattr(val, "detach") else np.asarray(val)
                    conv.append(_VecArray(arr) if ann.is_vec else arr)
            elif ann in (float, np.float32):
                conv.append(np.float32(val))
            elif ann in (int, np.int32):
                conv.append(int(val))
            else:
                conv.append(val)
        self._compiled(*conv)
        seeds = {}
        for w in taped:
            w.seeds(seeds)
        adj = tape.backward(seeds)
        for w in taped:
            w.collect(adj)


def kernel(fn):
    return _Kernel(fn)


Vector_cls = Vector


class _VectorFactory:
    """ti.Vector([...]) call syntax + isinstance support."""

    def __call__(self, data, dt=None):
        return Vector_cls(data, dt)


# `ti.Vector([a, b])` -> Vector
_vf = _VectorFactory()


def __getattr__(name):
    if name == "Vector":
        return _vf
    raise AttributeError(name)
