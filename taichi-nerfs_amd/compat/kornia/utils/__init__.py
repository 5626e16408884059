from . import grid  # noqa: F401
