"""Mirror of reference modules/hash_encoder.py: fp32 multiresolution hash-grid `HashEncoder` (:147-285).

The forward gather and the scatter-add backward are the gfx950 kernels ngp_hash_fwd_f32 / ngp_hash_bwd_f32.
The backward is the TRUE gradient of the forward w.r.t. the table (the reference hands autograd a tensor that
torch then adds to itself, i.e. 2x the gradient -- SURVEY.md H7; Adam is invariant to that factor)."""
import torch

from ngp_hip import ops as _ops
from .utils import scale_in_level_np


class _HashEncodeF32(torch.autograd.Function):

    @staticmethod
    def forward(ctx, positions, table, levels):
        ctx.levels = levels
        ctx.save_for_backward(positions)
        ctx.table_numel = table.numel()
        return _ops.hash_fwd_f32(positions, table, levels)

    @staticmethod
    def backward(ctx, dout):
        (positions,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.table_numel, device=dout.device, dtype=torch.float32)
        _ops.hash_bwd_f32(positions, dout.contiguous().float(), ctx.levels, dtable)
        return None, dtable, None


class HashEncoder(torch.nn.Module):
    """positions [N,3] f32 in [0,1] -> embedding [N, levels*feature_per_level] f32 (level-major)."""

    def __init__(self, max_params: float = 2**19, levels: int = 16, base_res: float = 16.0, max_res: float = 2048.0,
                 feature_per_level: int = 2):
        super().__init__()
        levels = int(levels)
        self.log_b = scale_in_level_np(base_res=base_res, max_res=max_res, levels=levels)
        self.base_res = base_res
        self.hash_level = levels
        self.max_params = max_params
        self.feature_per_level = feature_per_level
        self.out_dim = feature_per_level * levels

        # level table: same host arithmetic as reference :183-205, done once by the C ABI helper
        self._levels = _ops.make_levels(max_params, levels, base_res, max_res, feature_per_level)
        _, _, sizes, offsets = _ops.levels_to_numpy(self._levels)
        self.register_buffer('offsets', torch.tensor(offsets.astype('int64'), dtype=torch.int32), persistent=False)
        self.register_buffer('hash_map_sizes', torch.tensor(sizes.astype('int64'), dtype=torch.int32), persistent=False)
        self.begin_fast_hash_level = int(self._levels.begin_fast_hash_level)
        self.total_param_size = int(self._levels.total_entries) * feature_per_level

        print(f'Hash Encoder: base_res={base_res} max_res={max_res} hash_level={levels} '
              f'feat_per_level={feature_per_level} per_level_scale={self.log_b} '
              f'total_hash_size={int(self._levels.total_entries)} ')

        self.hash_table = torch.nn.Parameter(torch.zeros(self.total_param_size, dtype=torch.float32), requires_grad=True)
        torch.nn.init.uniform_(self.hash_table)          # U(0,1) like reference :227

    @property
    def levels_struct(self):
        return self._levels

    def forward(self, positions):
        return _HashEncodeF32.apply(positions.contiguous(), self.hash_table.contiguous(), self._levels)
