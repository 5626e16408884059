"""Mirror of reference modules/rendering.py: `render()` (:12-57) with its train (:161-228) and progressive
test-time (:62-158) paths, on the HIP operators."""
import os

import torch

from ngp_hip import ops as _ops
from .intersection import ray_aabb_intersection
from .ray_march import raymarching_test, raymarching_train
from .volume_render_test import composite_test

MAX_SAMPLES = 1024
NEAR_DISTANCE = 0.01


class TrainResults(dict):
    """Result dictionary of the fused training render.  The reference returns `deltas`, `ts`, `ws` as fresh [S] tensors
    (S = rm_samples, rendering.py:181-215); the fused render keeps them in its N*MAX_SAMPLES-row arena and never reads S back
    to the host.  Reading one of the three through the mapping protocol materialises exactly what the reference returns --
    a [S] copy (one host read of S, paid only by a caller that asks; `ws` stays attached to the autograd graph) -- which the
    next render() can no longer overwrite.  This package's own distortion loss reads the arena rows in place (`padded()`):
    `rays_a` addresses them identically, so train.py with --distortion_loss_w > 0 runs without the host read."""

    def __init__(self, data, padded, arena=None, lazy=None):
        super().__init__(data)
        self._padded = padded
        self._lazy = dict(lazy or {})        # key -> thunk: small reductions only a log line reads (rm_samples, vr_samples)
        self._arena, self._generation = arena, (arena.generation if arena is not None else None)

    def padded(self, key):
        """The arena-backed tensor ([N*MAX_SAMPLES] rows, live rows [0, rm_samples)); valid until the next training render."""
        return self._padded[key]

    def __missing__(self, key):
        if key in self._lazy:
            value = self._lazy.pop(key)()
            self[key] = value
            return value
        if key not in self._padded:
            raise KeyError(key)
        if self._arena is not None and self._arena.generation != self._generation:
            raise RuntimeError("render(): results[%r] of a training render was first read after a later training render with the same ray "
                               "count had reused its sample arena; read it before the next render(), or set NGP_FUSED_RENDER=0 for "
                               "the reference's per-call buffers" % key)
        n_live = int(self['rm_samples'])
        value = self._padded[key][:n_live].clone()
        self[key] = value
        return value

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._padded or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return (list(dict.keys(self)) + [k for k in self._lazy if not dict.__contains__(self, k)]
                + [k for k in self._padded if not dict.__contains__(self, k)])

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    # the rest of the mapping protocol sees the lazily materialised keys too: dict's own pop / setdefault / copy
    # bypass __missing__
    def pop(self, key, *default):
        if key in self:
            value = self[key]
            dict.pop(self, key, None)
            self._padded.pop(key, None)
            self._lazy.pop(key, None)
            return value
        if default:
            return default[0]
        raise KeyError(key)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default

    def copy(self):
        """A plain dict with every key materialised (what `dict(results)` gives)."""
        return {k: self[k] for k in self.keys()}

    def __reduce__(self):
        return (dict, (self.copy(),))                  # pickling / deepcopy: the reference's plain dictionary


def _background(exp_step_factor, device):
    # synthetic scenes (exp_step_factor == 0) are composited over white, real scenes over black
    return torch.ones(3, device=device) if exp_step_factor == 0 else torch.zeros(3, device=device)


def render(model, rays_o, rays_d, test_time=False, exp_step_factor=0, T_threshold=1e-4, max_samples=MAX_SAMPLES):
    """rays_o, rays_d: [N,3].  Returns the reference's result dictionary (rgb, depth, opacity, ...)."""
    rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()      # geometry is fp32 (ray_utils.py:50)
    if not test_time and getattr(model, 'fused_train_ok', None) is not None and model.fused_train_ok(rays_o):
        # the fused training render does the slab test of intersection.py:22-37 inside its march launch (same arithmetic, same hits_t)
        return _render_rays_train(model, rays_o, rays_d, None, exp_step_factor, T_threshold)
    hits_t = ray_aabb_intersection(rays_o, rays_d, model.scale)
    if test_time:
        if rays_o.is_cuda and os.environ.get("NGP_FUSED_EVAL", "1") != "0":
            return _render_rays_test_oneshot(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold, max_samples)
        return _render_rays_test(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold, max_samples)
    return _render_rays_train(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold)


@torch.no_grad()
def _render_rays_test(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold, max_samples):
    """Progressive marching: every round each alive ray advances by up to N_samples occupied steps, the samples
    are shaded and composited in place, converged rays drop out and the per-round budget grows."""
    n_rays = len(rays_o)
    device = rays_o.device
    opacity = torch.zeros(n_rays, device=device)
    depth = torch.zeros(n_rays, device=device)
    rgb = torch.zeros(n_rays, 3, device=device)
    alive = torch.arange(n_rays, device=device)
    min_samples = 1 if exp_step_factor == 0 else 4
    marched = 0
    total_samples = 0
    while marched < max_samples:
        n_alive = len(alive)
        if n_alive == 0:
            break
        n_step = max(min(n_rays // n_alive, 64), min_samples)
        marched += n_step
        pack_info, ray_indices, deltas, ts = raymarching_test(
            rays_o, rays_d, hits_t, alive, model.density_bitfield, model.cascades, model.scale, exp_step_factor,
            model.grid_size, n_step)
        if ray_indices.shape[0] == 0:
            break
        o = rays_o[ray_indices, :3]
        d = rays_d[ray_indices, :3]
        sigmas, rgbs = model(o + ts[:, None] * d, d)
        composite_test(sigmas, rgbs, deltas, ts, pack_info, alive, T_threshold, opacity, depth, rgb)
        alive = alive[alive >= 0]
        total_samples += pack_info[:, 1].sum()
    rgb += _background(exp_step_factor, device) * (1 - opacity)[:, None]
    return {'opacity': opacity, 'depth': depth, 'rgb': rgb, 'total_samples': total_samples}


EVAL_CHUNK = 65536


@torch.no_grad()
def _render_rays_test_oneshot(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold, max_samples, chunk=None):
    """Evaluation render without the progressive rounds: the test-time march (no jitter) visits exactly the samples the
    training march visits with zero noise, and compositing a ray front to back until T <= T_threshold gives what the
    round-by-round `composite_test` accumulates (tests/test_gpu_configs.py::test_garden_eval_path_..., test_gpu_eval.py).
    So an image is rendered in chunks of 65536 rays as march -> shade -> composite, one host read of the sample count per
    chunk instead of two per round -- the reference's loop (kept below as `_render_rays_test`, NGP_FUSED_EVAL=0) needs
    hundreds of rounds for rays that never saturate.  `total_samples` counts the samples in front of the termination point,
    like the reference's sum of per-round counts up to the round granularity."""
    n_rays = len(rays_o)
    device = rays_o.device
    chunk = chunk or EVAL_CHUNK
    opacity = torch.empty(n_rays, device=device)
    depth = torch.empty(n_rays, device=device)
    rgb = torch.empty(n_rays, 3, device=device)
    total_samples = torch.zeros((), device=device, dtype=torch.int64)
    for a in range(0, n_rays, chunk):
        b = min(a + chunk, n_rays)
        o, d, h = rays_o[a:b], rays_d[a:b], hits_t[a:b].contiguous()
        noise = torch.zeros(b - a, device=device)
        rays_a, xyzs, dirs, deltas, ts, total = _ops.march_train(o, d, h, model.density_bitfield, noise, model.cascades, model.scale,
                                                                 exp_step_factor, model.grid_size, max_samples)
        if xyzs.shape[0] == 0:
            opacity[a:b] = 0; depth[a:b] = 0; rgb[a:b] = 0
            continue
        sigmas, rgbs = model(xyzs, dirs)
        vr, op_c, dep_c, rgb_c, _ws = _ops.composite_train_fwd(sigmas.contiguous().float(), rgbs.contiguous(), deltas, ts, rays_a,
                                                               T_threshold)
        opacity[a:b] = op_c; depth[a:b] = dep_c; rgb[a:b] = rgb_c
        total_samples += vr.sum()
    rgb += _background(exp_step_factor, device) * (1 - opacity)[:, None]
    return {'opacity': opacity, 'depth': depth, 'rgb': rgb, 'total_samples': total_samples}


def _render_rays_train(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold):
    """march (occupancy-skipping, jittered) -> shade -> differentiable front-to-back compositing."""
    if getattr(model, 'fused_train_ok', None) is not None and model.fused_train_ok(rays_o):
        # one autograd node, no host sync; the per-sample outputs stay in the N*MAX_SAMPLES arena until somebody reads them
        # through the result dictionary (TrainResults: reference-shaped [S] tensors on first access)
        from ngp_hip.fused import FusedTrainRender, RenderConfig, TrainArena
        cfg = RenderConfig.cached(model, exp_step_factor, T_threshold, MAX_SAMPLES)
        # (the background blend of :219-226 is part of the node; rm_samples / vr_samples are read by train.py's log lines only:
        # the two reductions are formed when somebody asks)
        rgb, opacity, depth, ws, total, vr_per_ray, rays_a = FusedTrainRender.apply(
            rays_o, rays_d, hits_t, model.pos_encoder.hash_table, *model._mlp_weights(), cfg)
        A = TrainArena.get(rays_o.device, rays_o.shape[0], MAX_SAMPLES)
        return TrainResults({'opacity': opacity, 'depth': depth, 'rgb': rgb, 'rays_a': rays_a},
                            {'deltas': A.deltas, 'ts': A.ts, 'ws': ws}, A,
                            lazy={'rm_samples': lambda: total[0], 'vr_samples': lambda: vr_per_ray.sum()})
    if hits_t is None:
        hits_t = ray_aabb_intersection(rays_o, rays_d, model.scale)
    rays_a, xyzs, dirs, deltas, ts, rm_samples = raymarching_train(
        rays_o, rays_d, hits_t, model.density_bitfield, model.cascades, model.scale, exp_step_factor, model.grid_size,
        MAX_SAMPLES)
    sigmas, rgbs = model(xyzs, dirs)
    vr_samples, opacity, depth, rgb, ws = model.render_func(sigmas, rgbs, deltas, ts, rays_a, T_threshold)
    rgb = rgb + _background(exp_step_factor, rays_o.device) * (1 - opacity)[:, None]
    return {'deltas': deltas, 'ts': ts, 'rm_samples': rm_samples, 'vr_samples': vr_samples, 'opacity': opacity,
            'depth': depth, 'rgb': rgb, 'ws': ws, 'rays_a': rays_a}
