"""Mirror of reference modules/hash_encoder_half.py: the half2 `HashEncoder` (:218-368).

fp32 master table [entries, F] -> fp16 copy per forward (:367); f16 gather with f16 accumulation
(ngp_hash_fwd_f16); explicit backward with one packed f16x2 atomic per corner (ngp_hash_bwd_f16,
global_atomic_pk_add_f16).  The gradient buffer is a genuine fp16 buffer (the reference binds an fp32 tensor to
an f16-typed ndarray) and is returned to autograd as fp32 like the parameter."""
import torch

from ngp_hip import ops as _ops
from .utils import scale_in_level_np


class _HashEncodeF16(torch.autograd.Function):

    @staticmethod
    def forward(ctx, positions, table_h, module):
        ctx.module = module
        ctx.save_for_backward(positions)
        return _ops.hash_fwd_f16(positions, table_h, module._levels)

    @staticmethod
    def backward(ctx, dout):
        (positions,) = ctx.saved_tensors
        m = ctx.module
        grad_h = m._grad_f16(dout.device)
        grad_h.zero_()                                                   # reference :350-352
        _ops.hash_bwd_f16(positions, dout.contiguous().to(torch.float16), m._levels, grad_h)
        return None, grad_h, None


class HashEncoder(torch.nn.Module):

    def __init__(self, max_params: float = 2**19, levels: int = 16, base_res: float = 16.0, max_res: float = 2048.0,
                 feature_per_level: int = 2):
        super().__init__()
        levels = int(levels)
        if feature_per_level != 2:
            raise NotImplementedError("the half2 encoder packs exactly 2 features per entry")
        self.log_b = scale_in_level_np(base_res=base_res, max_res=max_res, levels=levels)
        self.base_res = base_res
        self.hash_level = levels
        self.max_params = max_params
        self.feature_per_level = feature_per_level
        self.out_dim = feature_per_level * levels

        self._levels = _ops.make_levels(max_params, levels, base_res, max_res, feature_per_level)
        _, _, sizes, offsets = _ops.levels_to_numpy(self._levels)
        self.register_buffer('offsets', torch.tensor(offsets.astype('int64'), dtype=torch.int32), persistent=False)
        self.register_buffer('hash_map_sizes', torch.tensor(sizes.astype('int64'), dtype=torch.int32), persistent=False)
        self.begin_fast_hash_level = int(self._levels.begin_fast_hash_level)
        entries = int(self._levels.total_entries)
        self.total_param_size = entries * feature_per_level

        print(f'Hash Encoder: base_res={base_res} max_res={max_res} hash_level={levels} '
              f'feat_per_level={feature_per_level} per_level_scale={self.log_b} total_hash_size={entries} ')

        self.hash_table = torch.nn.Parameter(torch.zeros(entries, feature_per_level, dtype=torch.float32),
                                             requires_grad=True)
        torch.nn.init.uniform_(self.hash_table, -1e-4, 1e-4)               # reference :299
        # persistent like the reference (:300-306) so checkpoints interchange
        self.register_buffer('hash_grad', torch.zeros_like(self.hash_table, dtype=torch.float32))
        self._grad_h = None

    @property
    def levels_struct(self):
        return self._levels

    def table_f16(self):
        """fp16 copy of the fp32 master table for the fused kernels; re-cast (the reference casts on every forward, :367) only
        when the parameter was written through torch.  FusedTrainer updates parameter AND copy in its Adam kernel."""
        t = self.hash_table
        ver = (t.data_ptr(), t._version)
        if getattr(self, "_f16", None) is None or self._f16.device != t.device:
            self._f16, self._f16_ver = torch.empty(t.shape, device=t.device, dtype=torch.float16), None
        if self._f16_ver != ver:
            self._f16.copy_(t.detach())
            self._f16_ver = ver
        return self._f16

    def _grad_f16(self, device):
        if self._grad_h is None or self._grad_h.device != device:
            self._grad_h = torch.zeros(self.hash_table.shape, device=device, dtype=torch.float16)
        return self._grad_h

    def forward(self, positions):
        table_h = self.hash_table.to(torch.float16).contiguous()           # reference :367
        return _HashEncodeF16.apply(positions.contiguous(), table_h, self).view(-1, self.out_dim)
