"""Mirror of reference modules/volume_render_test.py: in-place test-time compositing `composite_test` (:4-54)."""
import torch

from ngp_hip import ops as _ops


def composite_test(sigmas, rgbs, deltas, ts, pack_info, alive_indices, T_threshold, opacity, depth, rgb):
    """Accumulates into opacity/depth/rgb (indexed by ray) and marks converged / sample-less rays with
    alive_indices[n] = -1, all in place."""
    _ops.composite_test(sigmas.contiguous().float(), rgbs.contiguous(), deltas.contiguous(), ts.contiguous(),
                        pack_info.contiguous().to(torch.int64), alive_indices, T_threshold, opacity, depth, rgb)
