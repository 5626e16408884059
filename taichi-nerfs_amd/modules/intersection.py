"""Mirror of reference modules/intersection.py (ray_aabb_intersect :8-37, ray_aabb_intersection :40-55)."""
import ctypes

import torch

from ngp_hip import ops as _ops
from .utils import NEAR_DISTANCE  # noqa: F401  (re-exported like the reference)


def ray_aabb_intersect(hits_t, rays_o, rays_d, scale):
    """Kernel-style entry (fills a caller-provided hits_t), kept because deployment code imports it."""
    hits_t.copy_(_ops.ray_aabb(rays_o.contiguous(), rays_d.contiguous(), scale))


def ray_aabb_intersection(rays_o, rays_d, scale):
    """rays_o, rays_d: [N,3] f32 -> hits_t [N,2] = (max(t_near, 0.01), t_far) or (-1,-1) on a miss."""
    return _ops.ray_aabb(rays_o.contiguous(), rays_d.contiguous(), scale)
