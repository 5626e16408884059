"""No-op stand-in for the `taichi` package: the driver only calls ti.init()/ti.reset() (train.py:27-32,308) and gui.py
decorates one copy kernel at import time.  All kernels of the hot path are HIP (libngp_hip.so)."""
import types as _types

cuda = "cuda"
cpu = "cpu"
vulkan = "vulkan"
f32, f16, i32, u32, u8 = "f32", "f16", "i32", "u32", "u8"


def init(*args, **kwargs):
    return None


def reset():
    return None


def sync():
    return None


def kernel(fn):
    def _unavailable(*a, **k):
        raise RuntimeError("Taichi kernels are not available: this build runs the HIP kernels of libngp_hip.so")
    _unavailable.__name__ = getattr(fn, "__name__", "kernel")
    return _unavailable


def func(fn):
    return fn


def template():
    return None


def static(x):
    return x


def ndrange(*a):
    import itertools
    return itertools.product(*[range(int(x)) for x in a])


class _Types:
    @staticmethod
    def ndarray(*a, **k):
        return None

    @staticmethod
    def vector(*a, **k):
        return None


types = _Types()
math = _types.ModuleType("taichi.math")
ui = _types.ModuleType("taichi.ui")
