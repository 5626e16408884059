"""Mirror of reference modules/rendering.py: `render()` (:12-57) with its train (:161-228) and progressive
test-time (:62-158) paths, on the HIP operators."""
import torch

from .intersection import ray_aabb_intersection
from .ray_march import raymarching_test, raymarching_train
from .volume_render_test import composite_test

MAX_SAMPLES = 1024
NEAR_DISTANCE = 0.01


def _background(exp_step_factor, device):
    # synthetic scenes (exp_step_factor == 0) are composited over white, real scenes over black
    return torch.ones(3, device=device) if exp_step_factor == 0 else torch.zeros(3, device=device)


def render(model, rays_o, rays_d, test_time=False, exp_step_factor=0, T_threshold=1e-4, max_samples=MAX_SAMPLES):
    """rays_o, rays_d: [N,3].  Returns the reference's result dictionary (rgb, depth, opacity, ...)."""
    rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()      # geometry is fp32 (ray_utils.py:50)
    hits_t = ray_aabb_intersection(rays_o, rays_d, model.scale)
    if test_time:
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


def _render_rays_train(model, rays_o, rays_d, hits_t, exp_step_factor, T_threshold):
    """march (occupancy-skipping, jittered) -> shade -> differentiable front-to-back compositing."""
    if getattr(model, 'fused_train_ok', None) is not None and model.fused_train_ok(rays_o):
        # one autograd node, no host sync; deltas / ts / ws come back padded to the N*MAX_SAMPLES arena
        # (only rows [0, rm_samples) are live -- rays_a addresses them exactly like the reference)
        from ngp_hip.fused import FusedTrainRender, RenderConfig
        cfg = RenderConfig(model, exp_step_factor, T_threshold, MAX_SAMPLES)
        rgb, opacity, depth, ws, rm_samples, vr_samples, rays_a = FusedTrainRender.apply(
            rays_o.contiguous().float(), rays_d.contiguous().float(), hits_t, model.pos_encoder.hash_table,
            *model._mlp_weights(), cfg)
        rgb = rgb + _background(exp_step_factor, rays_o.device) * (1 - opacity)[:, None]
        from ngp_hip.fused import TrainArena
        A = TrainArena.get(rays_o.device, rays_o.shape[0], MAX_SAMPLES)
        return {'deltas': A.deltas, 'ts': A.ts, 'rm_samples': rm_samples, 'vr_samples': vr_samples, 'opacity': opacity,
                'depth': depth, 'rgb': rgb, 'ws': ws, 'rays_a': rays_a}
    rays_a, xyzs, dirs, deltas, ts, rm_samples = raymarching_train(
        rays_o, rays_d, hits_t, model.density_bitfield, model.cascades, model.scale, exp_step_factor, model.grid_size,
        MAX_SAMPLES)
    sigmas, rgbs = model(xyzs, dirs)
    vr_samples, opacity, depth, rgb, ws = model.render_func(sigmas, rgbs, deltas, ts, rays_a, T_threshold)
    rgb = rgb + _background(exp_step_factor, rays_o.device) * (1 - opacity)[:, None]
    return {'deltas': deltas, 'ts': ts, 'rm_samples': rm_samples, 'vr_samples': vr_samples, 'opacity': opacity,
            'depth': depth, 'rgb': rgb, 'ws': ws, 'rays_a': rays_a}
