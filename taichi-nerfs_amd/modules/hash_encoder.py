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


class _HashEncodeBF16(torch.autograd.Function):
    """bf16-stored copy of the fp32 master table in the forward (half the gather bytes), the same fp32 scatter-add
    backward: d(encoding)/d(table) does not depend on the table values."""

    @staticmethod
    def forward(ctx, positions, table, table_bf16, levels):
        ctx.levels = levels
        ctx.save_for_backward(positions)
        ctx.table_numel = table.numel()
        return _ops.hash_fwd_bf16(positions, table_bf16, levels)

    @staticmethod
    def backward(ctx, dout):
        (positions,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.table_numel, device=dout.device, dtype=torch.float32)
        _ops.hash_bwd_f32(positions, dout.contiguous().float(), ctx.levels, dtable)
        return None, dtable, None, None


class HashEncoder(torch.nn.Module):
    """positions [N,3] f32 in [0,1] -> embedding [N, levels*feature_per_level] f32 (level-major).

    table_dtype=torch.bfloat16 (not in the reference; BASELINE config 2 names a bf16 hash grid): the forward gathers from
    a bf16 copy of the fp32 master `hash_table` (refreshed whenever the parameter changes, the way hash_encoder_half.py:367
    re-casts its fp16 copy every call); parameter, gradient, optimizer state and state_dict stay fp32."""

    def __init__(self, max_params: float = 2**19, levels: int = 16, base_res: float = 16.0, max_res: float = 2048.0,
                 feature_per_level: int = 2, table_dtype=None):
        super().__init__()
        if table_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("table_dtype must be None / torch.float32 / torch.bfloat16")
        if table_dtype == torch.bfloat16 and feature_per_level != 2:
            raise ValueError("the bf16 table packs feature pairs: feature_per_level must be 2")
        self.table_dtype = torch.bfloat16 if table_dtype == torch.bfloat16 else torch.float32
        self._bf16, self._bf16_ver = None, None
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

    def table_bf16(self):
        """The bf16 copy the forward gathers from; re-cast only when the parameter was written through torch (optimizer
        step, load_state_dict -- its version counter moves).  FusedTrainer updates parameter AND copy in its Adam kernel."""
        t = self.hash_table
        ver = (t.data_ptr(), t._version)
        if self._bf16 is None or self._bf16.device != t.device:
            self._bf16, self._bf16_ver = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16), None
        if self._bf16_ver != ver:
            _ops.cast_bf16(t.detach(), self._bf16)
            self._bf16_ver = ver
        return self._bf16

    def forward(self, positions):
        if self.table_dtype == torch.bfloat16:
            return _HashEncodeBF16.apply(positions.contiguous(), self.hash_table, self.table_bf16(), self._levels)
        return _HashEncodeF32.apply(positions.contiguous(), self.hash_table.contiguous(), self._levels)
