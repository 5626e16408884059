"""MI355X-native Instant-NGP hot path: ctypes binding (lib), tensor launchers (ops)."""
from . import lib, ops  # noqa: F401
