"""`from taichi.math import vec3, vec2, uvec3` support for the shim."""
from . import math as _m

vec2 = _m.vec2
vec3 = _m.vec3
uvec3 = _m.uvec3
ivec3 = _m.ivec3
clamp = _m.clamp
sign = _m.sign
pow = _m.pow  # noqa: A001
