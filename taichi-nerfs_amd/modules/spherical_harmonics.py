"""Mirror of reference modules/spherical_harmonics.py: degree-4 (16 coefficient) `DirEncoder` (:62-102).
The polynomials are evaluated on the input as given (callers pass (d+1)/2, reference networks.py:163)."""
import torch

from ngp_hip import ops as _ops


class _SH16(torch.autograd.Function):

    @staticmethod
    def forward(ctx, dirs):
        ctx.save_for_backward(dirs)
        return _ops.sh16_fwd(dirs)

    @staticmethod
    def backward(ctx, dout):
        (dirs,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None
        return _ops.sh16_bwd(dirs, dout.contiguous().float())


class DirEncoder(torch.nn.Module):

    def __init__(self):
        super().__init__()
        self.out_dim = 16

    def forward(self, dirs):
        return _SH16.apply(dirs.contiguous().float())
