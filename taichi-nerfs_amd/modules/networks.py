"""Mirror of reference modules/networks.py: `NGP` (:33-290), `MLP` (:293-380), `TruncExp` (:18-30).

Same constructor arguments, attributes, buffers and state_dict keys (pos_encoder.hash_table,
xyz_encoder.hidden_layers.N.weight, ..., density_grid, density_bitfield, grid_coords) so reference checkpoints
load both ways.  The occupancy-grid maintenance is the reference's algorithm on the HIP kernels."""
from typing import Callable, Optional

import numpy as np
import torch
from torch import nn

import os

from ngp_hip import ops as _ops
from .rendering import NEAR_DISTANCE
from .spherical_harmonics import DirEncoder
from .utils import morton3D, morton3D_invert, packbits
from .volume_train import VolumeRenderer


class TruncExp(torch.autograd.Function):
    """exp() in fp32 with the backward exponent clamped to [-15, 15] (reference :18-30)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


class _FusedShade(torch.autograd.Function):
    """xyz_encoder + TruncExp + direction normalisation + SH16 + rgb_net in one MFMA kernel each way
    (ngp_mlp_fwd / ngp_mlp_bwd).  Numerically this is the autocast(fp16) formulation of reference
    networks.py:136-166: fp16 operands and layer outputs, fp32 accumulation, fp32 exp."""

    @staticmethod
    def forward(ctx, enc, dirs, w1, w2, w3, w4, w5):
        wpack = _ops.mlp_pack((w1, w2, w3, w4, w5))
        sigmas, rgbs = _ops.mlp_fwd(enc, dirs, wpack)
        ctx.save_for_backward(enc, dirs, wpack)
        ctx.set_materialize_grads(False)
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, g_sigmas, g_rgbs):
        enc, dirs, wpack = ctx.saved_tensors
        n = enc.shape[0]
        g_sigmas = torch.zeros(n, device=enc.device) if g_sigmas is None else g_sigmas.contiguous().float()
        g_rgbs = torch.zeros(n, 3, device=enc.device, dtype=torch.float16) if g_rgbs is None \
            else g_rgbs.contiguous().to(torch.float16)
        d_enc, dW = _ops.mlp_bwd(enc, dirs, wpack, g_sigmas, g_rgbs)
        grads = [g.view(shape) for g, shape in zip(dW.split(_ops.MLP_SPLITS), _ops.MLP_SHAPES)]
        return (d_enc, None, *grads)


def _cell_coords(grid_size):
    """All G^3 integer cell coordinates, [G^3, 3] int32 (any enumeration order serves: cells are addressed by
    their Morton code)."""
    r = torch.arange(grid_size, dtype=torch.int32)
    return torch.stack(torch.meshgrid(r, r, r, indexing='ij'), dim=-1).reshape(-1, 3)


class NGP(nn.Module):

    def __init__(self, scale: float = 0.5, pos_encoder_type: str = 'hash', levels: int = 16, feature_per_level: int = 2,
                 log2_T: int = 19, base_res: int = 16, max_res: int = 2048, half_opt: bool = False,
                 xyz_net_width: int = 64, xyz_net_depth: int = 1, xyz_net_out_dim: int = 16, rgb_net_depth: int = 2,
                 rgb_net_width: int = 64, table_dtype=None):
        super().__init__()
        self.scale = scale
        self.half_opt = bool(half_opt)
        self.register_buffer('center', torch.zeros(1, 3))
        self.register_buffer('xyz_min', -torch.ones(1, 3) * scale)
        self.register_buffer('xyz_max', torch.ones(1, 3) * scale)
        self.register_buffer('half_size', (self.xyz_max - self.xyz_min) / 2)

        # cascade k covers [-2^(k-1), 2^(k-1)]^3
        self.cascades = max(1 + int(np.ceil(np.log2(2 * scale))), 1)
        self.grid_size = 128
        G3 = self.grid_size**3
        self.register_buffer('density_bitfield', torch.zeros(self.cascades * G3 // 8, dtype=torch.uint8))
        self.register_buffer('density_grid', torch.zeros(self.cascades, G3))
        self.register_buffer('grid_coords', _cell_coords(self.grid_size))

        if pos_encoder_type != 'hash':
            raise NotImplementedError("only the hash-grid position encoder is on the MI355X hot path "
                                      "(the reference's tri-plane encoder is out of scope, see DESIGN.md)")
        if half_opt:
            from .hash_encoder_half import HashEncoder
        else:
            from .hash_encoder import HashEncoder
        enc_kw = {}
        if table_dtype is not None:                  # bf16 storage copy of the fp32 table (fp32 encoder only)
            if half_opt:
                raise ValueError("table_dtype applies to the fp32 encoder; half_opt already selects the fp16 table")
            enc_kw['table_dtype'] = table_dtype
        self.pos_encoder = HashEncoder(max_params=2**log2_T, base_res=base_res, max_res=max_res, levels=levels,
                                       feature_per_level=feature_per_level, **enc_kw)

        self.xyz_encoder = MLP(input_dim=self.pos_encoder.out_dim, output_dim=xyz_net_out_dim, net_depth=xyz_net_depth,
                               net_width=xyz_net_width, bias_enabled=False)
        self.dir_encoder = DirEncoder()
        self.rgb_net = MLP(input_dim=self.dir_encoder.out_dim + self.xyz_encoder.output_dim, output_dim=3,
                           net_depth=rgb_net_depth, net_width=rgb_net_width, bias_enabled=False,
                           output_activation=nn.Sigmoid())
        self.render_func = VolumeRenderer()
        # the fused MFMA MLP covers exactly the default architecture; anything else runs the torch layers
        self.use_fused_mlp = (os.environ.get("NGP_FUSED_MLP", "1") != "0" and self.pos_encoder.out_dim == 32
                              and xyz_net_width == 64 and xyz_net_depth == 1 and xyz_net_out_dim == 16
                              and rgb_net_depth == 2 and rgb_net_width == 64)

    def _mlp_weights(self):
        return (self.xyz_encoder.hidden_layers[0].weight, self.xyz_encoder.output_layer.weight,
                self.rgb_net.hidden_layers[0].weight, self.rgb_net.hidden_layers[1].weight, self.rgb_net.output_layer.weight)

    def fused_train_ok(self, rays):
        """Whole-render fusion (ngp_hip/fused.py): fp32 (or bf16-copy) hash table or the half2 encoder + default MLPs +
        autocast(fp16) numerics."""
        return (self._fused_ok(rays) and torch.is_grad_enabled() and os.environ.get("NGP_FUSED_RENDER", "1") != "0")

    def _fused_ok(self, x):
        """Fused path = the fp16-autocast numerics of the reference's training/eval loops (train.py:177,250)."""
        return self.use_fused_mlp and x.is_cuda and torch.is_autocast_enabled()

    # ------------------------------------------------------------------------------------------ shading
    def density(self, x, return_feat=False):
        """x: [N,3] in [-scale, scale] -> sigmas [N] (and the 16-wide geometry feature)."""
        x = (x - self.xyz_min) / (self.xyz_max - self.xyz_min)
        if not return_feat and not torch.is_grad_enabled() and self._fused_ok(x):
            enc = self.pos_encoder(x).float().contiguous()
            return _ops.mlp_density(enc, _ops.mlp_pack(self._mlp_weights()))
        h = self.xyz_encoder(self.pos_encoder(x))
        sigmas = TruncExp.apply(h[:, 0])
        return (sigmas, h) if return_feat else sigmas

    def forward(self, x, d):
        """x: [N,3] positions, d: [N,3] directions -> (sigmas [N], rgbs [N,3])."""
        if self._fused_ok(x):
            x01 = (x - self.xyz_min) / (self.xyz_max - self.xyz_min)
            enc = self.pos_encoder(x01).float().contiguous()
            return _FusedShade.apply(enc, d.contiguous().float(), *self._mlp_weights())
        sigmas, h = self.density(x, return_feat=True)
        d = d / torch.norm(d, dim=1, keepdim=True)
        sh = self.dir_encoder((d + 1) / 2)
        rgbs = self.rgb_net(torch.cat([sh, h], 1))
        return sigmas, rgbs

    # ------------------------------------------------------------------------------------------ occupancy grid
    @torch.no_grad()
    def get_all_cells(self):
        """Every cell of every cascade, enumerated in Morton order (any enumeration addresses the same cells; this one
        makes consecutive encoder queries spatially coherent: the 2 M-point warm-up encode runs 1.6x faster)."""
        cached = getattr(self, '_all_cells_sorted', None)
        if cached is None or cached[0].device != self.grid_coords.device:
            indices = morton3D(self.grid_coords).long()
            order = torch.argsort(indices)
            cached = self._all_cells_sorted = (indices[order].contiguous(), self.grid_coords[order].contiguous())
        return [cached] * self.cascades

    @torch.no_grad()
    def sample_uniform_and_occupied_cells(self, M, density_threshold):
        """Per cascade: M uniformly random cells + M cells drawn from those above the threshold."""
        dev = self.density_grid.device
        cells = []
        for c in range(self.cascades):
            coords1 = torch.randint(self.grid_size, (M, 3), dtype=torch.int32, device=dev)
            indices1 = morton3D(coords1).long()
            indices2 = torch.nonzero(self.density_grid[c] > density_threshold)[:, 0]
            if len(indices2) > 0:
                indices2 = indices2[torch.randint(len(indices2), (M,), device=dev)]
            coords2 = morton3D_invert(indices2.int())
            indices, coords = torch.cat([indices1, indices2]), torch.cat([coords1, coords2])
            # bucket the (random) sample by its 8^3-cell block: same cells, same values, but neighbouring encoder queries
            # now share hash-grid lines (the 1 M-point encode is gather-bound; 900 -> 540 us for a 64 us 12-bit key sort)
            order = torch.sort((indices >> 9).to(torch.int16))[1]
            cells.append((indices[order], coords[order]))
        return cells

    @torch.no_grad()
    def mark_invisible_cells(self, K, poses, img_wh, chunk=32**3):
        """density_grid = -1 for cells no training camera sees (or that sit closer than NEAR_DISTANCE)."""
        n_cams = poses.shape[0]
        self.count_grid = torch.zeros_like(self.density_grid)
        w2c_R = poses[:, :3, :3].transpose(1, 2)
        w2c_T = -w2c_R @ poses[:, :3, 3:]
        cells = self.get_all_cells()
        for c in range(self.cascades):
            indices, coords = cells[c]
            s = min(2**(c - 1), self.scale)
            half_grid = s / self.grid_size
            for i in range(0, len(indices), chunk):
                xyzs = coords[i:i + chunk] / (self.grid_size - 1) * 2 - 1
                xyzs_w = (xyzs * (s - half_grid)).T
                uvd = K @ (w2c_R @ xyzs_w + w2c_T)
                uv = uvd[:, :2] / uvd[:, 2:]
                in_image = (uvd[:, 2] >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < img_wh[0]) & (uv[:, 1] >= 0) & (uv[:, 1] < img_wh[1])
                seen = (uvd[:, 2] >= NEAR_DISTANCE) & in_image
                self.count_grid[c, indices[i:i + chunk]] = count = seen.sum(0) / n_cams
                too_near = ((uvd[:, 2] < NEAR_DISTANCE) & in_image).any(0)
                valid = (count > 0) & (~too_near)
                self.density_grid[c, indices[i:i + chunk]] = torch.where(valid, 0., -1.)

    @torch.no_grad()
    def update_density_grid(self, density_threshold, warmup=False, decay=0.95, erode=False, jitter=None):
        """Reference signature (:255-259) plus `jitter` (tests only): callable (cascade, n) -> [n, 3] uniforms that replace the
        in-cell torch.rand_like draw of the warm-up form, row i belonging to the cell with Morton code i."""
        if jitter is not None and not warmup:
            raise ValueError("an explicit jitter is defined for the warm-up form (all cells) only")
        if (not erode and self.density_grid.is_cuda and self._fused_ok(self.density_grid)
                and os.environ.get("NGP_FUSED_OCCUPANCY", "1") != "0"):
            # same algorithm, device-resident (no torch.nonzero / .item() host round trips): ngp_hip/occupancy.py
            upd = getattr(self, '_occ_updater', None)
            if upd is None or upd.dev != self.density_grid.device:
                from ngp_hip.occupancy import OccupancyUpdater
                upd = self._occ_updater = OccupancyUpdater(self)
            if not self.density_grid.is_contiguous():
                self.density_grid = self.density_grid.contiguous()
            return upd.update(density_threshold, warmup=warmup, decay=decay, jitter=jitter)
        fresh = torch.zeros_like(self.density_grid)
        cells = self.get_all_cells() if warmup else \
            self.sample_uniform_and_occupied_cells(self.grid_size**3 // 4, density_threshold)
        for c in range(self.cascades):
            indices, coords = cells[c]
            s = min(2**(c - 1), self.scale)
            half_grid = s / self.grid_size
            xyzs_w = (coords / (self.grid_size - 1) * 2 - 1) * (s - half_grid)
            u = torch.rand_like(xyzs_w) if jitter is None else jitter(c, len(indices)).to(xyzs_w)   # (all cells: Morton-sorted rows)
            xyzs_w += (u * 2 - 1) * half_grid                                # random point inside the cell
            fresh[c, indices] = self.density(xyzs_w).float()
        if erode:
            decay = torch.clamp(decay**(1 / self.count_grid), 0.1, 0.95)
        self.density_grid = torch.where(self.density_grid < 0, self.density_grid,
                                        torch.maximum(self.density_grid * decay, fresh))
        mean_density = self.density_grid[self.density_grid > 0].mean().item()
        packbits(self.density_grid.reshape(-1).contiguous(), min(mean_density, density_threshold), self.density_bitfield)


class MLP(nn.Module):
    """Bias-optional ReLU MLP with the reference's constructor (:293-380) and parameter names
    (`hidden_layers.N`, `output_layer`)."""

    def __init__(self, input_dim: int, output_dim: int = None, net_depth: int = 8, net_width: int = 256, skip_layer: int = 4,
                 hidden_init: Callable = nn.init.xavier_uniform_, hidden_activation: Callable = nn.ReLU(),
                 output_enabled: bool = True, output_init: Optional[Callable] = nn.init.xavier_uniform_,
                 output_activation: Optional[Callable] = nn.Identity(), bias_enabled: bool = True,
                 bias_init: Callable = nn.init.zeros_):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.net_depth, self.net_width, self.skip_layer = net_depth, net_width, skip_layer
        self.hidden_init, self.hidden_activation = hidden_init, hidden_activation
        self.output_enabled, self.output_init, self.output_activation = output_enabled, output_init, output_activation
        self.bias_enabled, self.bias_init = bias_enabled, bias_init

        self.hidden_layers = nn.ModuleList()
        fan_in = input_dim
        for i in range(net_depth):
            self.hidden_layers.append(nn.Linear(fan_in, net_width, bias=bias_enabled))
            fan_in = net_width + input_dim if self._skips_at(i) else net_width
        if output_enabled:
            self.output_layer = nn.Linear(fan_in, output_dim, bias=bias_enabled)
        else:
            self.output_dim = fan_in
        self.initialize()

    def _skips_at(self, i):
        return self.skip_layer is not None and i % self.skip_layer == 0 and i > 0

    def _init_linear(self, layer, weight_init):
        if weight_init is not None:
            weight_init(layer.weight)
        if self.bias_enabled and self.bias_init is not None:
            self.bias_init(layer.bias)

    def initialize(self):
        for layer in self.hidden_layers:
            self._init_linear(layer, self.hidden_init)
        if self.output_enabled:
            self._init_linear(self.output_layer, self.output_init)

    def forward(self, x):
        skip_in = x
        for i, layer in enumerate(self.hidden_layers):
            x = self.hidden_activation(layer(x))
            if self._skips_at(i):
                x = torch.cat([x, skip_in], dim=-1)
        if self.output_enabled:
            x = self.output_activation(self.output_layer(x))
        return x


class VoxelGrid(nn.Module):
    """Placeholder so `from modules.networks import NGP, VoxelGrid, MODEL_DICT` (reference train.py:18) resolves.
    The reference's svox model is unfinished upstream (undefined names in its forward) and out of scope."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("model_name='svox' is not part of the MI355X hot path")


MODEL_DICT = {'ngp': NGP, 'svox': VoxelGrid}
