"""Mirror of reference modules/distortion.py: `distortion_loss(results)` (:8-12) and `DistortionLoss` (:122-194),
the Mip-NeRF-360 distortion regulariser (train.py:194-195, active when --distortion_loss_w > 0)."""
import torch

from ngp_hip import ops as _ops


def distortion_loss(results):
    """results: the dictionary render() returns in training mode. -> per-ray loss [N_rays]."""
    if hasattr(results, 'padded'):
        # fused training render (rendering.TrainResults): the per-sample rows are read in place from its arena -- rays_a addresses
        # them exactly like the reference's [S] tensors -- so no host read of the sample count is needed
        return DistortionLoss.apply(results.padded('ws'), results.padded('deltas'), results.padded('ts'), results['rays_a'])
    return DistortionLoss.apply(results['ws'], results['deltas'], results['ts'], results['rays_a'])


class DistortionLoss(torch.autograd.Function):
    """ws, deltas (interval lengths), ts (midpoints): per-sample [S]; rays_a [N,3] = (ray_idx, start, count).
    Differentiable w.r.t. ws only, like the reference."""

    @staticmethod
    def forward(ctx, ws, deltas, ts, rays_a):
        ws, deltas, ts, rays_a = ws.contiguous().float(), deltas.contiguous(), ts.contiguous(), rays_a.contiguous()
        loss, ws_inc, wts_inc = _ops.distortion_fwd(ws, deltas, ts, rays_a)
        ctx.save_for_backward(ws_inc, wts_inc, ws, deltas, ts, rays_a)
        return loss

    @staticmethod
    def backward(ctx, dL_dloss):
        ws_inc, wts_inc, ws, deltas, ts, rays_a = ctx.saved_tensors
        return _ops.distortion_bwd(dL_dloss.contiguous().float(), ws, deltas, ts, ws_inc, wts_inc, rays_a), None, None, None
