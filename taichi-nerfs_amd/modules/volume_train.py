"""Mirror of reference modules/volume_train.py: `VolumeRenderer` (:52-195).

forward(sigmas[S] f32, rgbs[S,3] f16|f32, deltas[S], ts[S], rays_a[N,3] i32, T_threshold)
  -> (vr_samples 0-d int tensor, opacity[N], depth[N], rgb[N,3], ws[S]); per-ray outputs are indexed by
  ray_idx = rays_a[:,0].  Differentiable w.r.t. sigmas and rgbs; the gradient of rgbs has rgbs' dtype."""
import torch

from ngp_hip import ops as _ops


class _Composite(torch.autograd.Function):

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        total_samples, opacity, depth, rgb, ws = _ops.composite_train_fwd(sigmas, rgbs, deltas, ts, rays_a, T_threshold)
        ctx.T_threshold = T_threshold
        ctx.save_for_backward(sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws)
        ctx.set_materialize_grads(False)
        vr = total_samples.sum()
        ctx.mark_non_differentiable(vr)
        return vr, opacity, depth, rgb, ws

    @staticmethod
    def backward(ctx, _g_total, g_opacity, g_depth, g_rgb, g_ws):
        sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws = ctx.saved_tensors
        n = rays_a.shape[0]

        def f32(g):
            return None if g is None else g.contiguous().float()

        g_rgb = f32(g_rgb)
        if g_rgb is None:
            g_rgb = torch.zeros(n, 3, device=sigmas.device, dtype=torch.float32)
        d_sigmas, d_rgbs = _ops.composite_train_bwd(f32(g_opacity), f32(g_depth), g_rgb, f32(g_ws), sigmas, rgbs, deltas, ts,
                                                    rays_a, opacity, depth, rgb, ws, ctx.T_threshold)
        return d_sigmas, d_rgbs, None, None, None, None


class VolumeRenderer(torch.nn.Module):

    def forward(self, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
        return _Composite.apply(sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous(), ts.contiguous(),
                                rays_a.contiguous(), T_threshold)
