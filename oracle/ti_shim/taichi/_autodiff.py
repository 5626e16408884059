"""Reverse-mode tape for ti_shim: emulates what `kernel.grad(...)` means for the three reference kernels whose backward exists
only inside Taichi's compiler (modules/hash_encoder.py:269, modules/spherical_harmonics.py:92, modules/volume_train.py:160).

TEST INFRASTRUCTURE ONLY.  The FORWARD source of the reference kernel is executed once more, with every torch tensor argument
that `requires_grad` wrapped in an ADArray: an element read before it was written is an INPUT leaf, an element the kernel
assigns is an OUTPUT whose seed is the tensor's `.grad` (what the reference's autograd glue stores there before it calls
`.grad`), and after the reverse sweep every input leaf's adjoint is ACCUMULATED into the input tensor's `.grad` -- Taichi's
torch-interop semantics.  Values and adjoints are IEEE binary32 throughout (one rounding per operation, like the interpreter's
forward); the order of the adjoint accumulations is the reverse of the (serial) forward order, which is A valid order, not
necessarily Taichi's: results agree with any other order to a few ulp of the largest partial sum.
Supported: + - * / unary minus, exp, min / max, comparisons (by value), casts to integer (value only: zero derivative)."""
import numpy as np

F = np.float32


class Tape:
    def __init__(self):
        self.parents = []          # per node: tuple of parent node ids
        self.partials = []         # per node: tuple of d(node)/d(parent) as float32

    def new(self, parents=(), partials=()):
        self.parents.append(parents)
        self.partials.append(partials)
        return len(self.parents) - 1

    def backward(self, seeds):
        adj = [F(0.0)] * len(self.parents)
        for node, s in seeds.items():
            adj[node] = F(adj[node] + F(s))
        for k in range(len(self.parents) - 1, -1, -1):
            a = adj[k]
            if a == 0:
                continue
            for p, d in zip(self.parents[k], self.partials[k]):
                adj[p] = F(adj[p] + F(a * d))
        return adj


class AD:
    """A taped float32 scalar."""
    __array_ufunc__ = None             # numpy scalars defer to __radd__ / __rmul__ ... instead of broadcasting over us
    __slots__ = ("t", "v", "i")

    def __init__(self, tape, v, i):
        self.t, self.v, self.i = tape, F(v), i

    @staticmethod
    def _val(o):
        return o.v if isinstance(o, AD) else F(o)

    def _mk(self, v, parents, partials):
        return AD(self.t, v, self.t.new(tuple(parents), tuple(F(d) for d in partials)))

    def _bin(self, o, v, da, db):
        if isinstance(o, AD):
            return self._mk(v, (self.i, o.i), (da, db))
        return self._mk(v, (self.i,), (da,))

    def __add__(self, o): return self._bin(o, F(self.v + self._val(o)), 1.0, 1.0)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, F(self.v - self._val(o)), 1.0, -1.0)
    def __rsub__(self, o): return self._mk(F(F(o) - self.v), (self.i,), (-1.0,))
    def __mul__(self, o): return self._bin(o, F(self.v * self._val(o)), self._val(o), self.v)
    __rmul__ = __mul__

    def __truediv__(self, o):
        b = self._val(o)
        return self._bin(o, F(self.v / b), F(F(1.0) / b), F(-F(self.v / b) / b))

    def __rtruediv__(self, o):
        a = F(o)
        return self._mk(F(a / self.v), (self.i,), (F(-F(a / self.v) / self.v),))

    def __neg__(self): return self._mk(F(-self.v), (self.i,), (-1.0,))
    def __pos__(self): return self

    def exp(self):
        e = F(np.exp(self.v))
        return self._mk(e, (self.i,), (e,))

    # comparisons and conversions act on the value
    def __lt__(self, o): return bool(self.v < self._val(o))
    def __le__(self, o): return bool(self.v <= self._val(o))
    def __gt__(self, o): return bool(self.v > self._val(o))
    def __ge__(self, o): return bool(self.v >= self._val(o))
    def __eq__(self, o): return bool(self.v == self._val(o))
    def __ne__(self, o): return bool(self.v != self._val(o))
    __hash__ = None
    def __float__(self): return float(self.v)
    def __int__(self): return int(self.v)
    def __bool__(self): return bool(self.v != 0)
    def __repr__(self): return "AD(%r)" % (self.v,)


def ad_min(a, b):
    return a if AD._val(a) <= AD._val(b) else b


def ad_max(a, b):
    return a if AD._val(a) >= AD._val(b) else b


class ADArray:
    """A torch tensor that requires grad, as a kernel argument under the tape."""

    def __init__(self, tape, tensor):
        self.tape, self.tensor = tape, tensor
        self.arr = tensor.detach().numpy()
        self.shape = self.arr.shape
        self.cells, self.leaves, self.written = {}, {}, set()

    @staticmethod
    def _key(idx):
        return tuple(int(i) for i in idx) if isinstance(idx, tuple) else (int(idx),)

    def __getitem__(self, idx):
        k = self._key(idx)
        c = self.cells.get(k)
        if c is None:
            c = AD(self.tape, self.arr[k], self.tape.new())
            self.cells[k], self.leaves[k] = c, c.i
        return c

    def __setitem__(self, idx, val):
        k = self._key(idx)
        self.written.add(k)
        if isinstance(val, AD):
            self.cells[k] = val
            self.arr[k] = val.v
        else:
            self.cells[k] = F(val)
            self.arr[k] = val

    def seeds(self, out):
        g = self.tensor.grad
        if g is None:
            return
        gn = g.detach().numpy()
        for k in self.written:
            c = self.cells[k]
            if isinstance(c, AD):
                out[c.i] = F(out.get(c.i, F(0.0)) + F(gn[k]))

    def collect(self, adj):
        if not self.leaves:
            return
        import torch
        if self.tensor.grad is None:
            self.tensor.grad = torch.zeros_like(self.tensor)
        gn = self.tensor.grad.detach().numpy()
        for k, node in self.leaves.items():
            gn[k] = F(gn[k] + adj[node])
