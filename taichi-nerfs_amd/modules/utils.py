"""Mirror of reference modules/utils.py: constants, level-table helpers and the three occupancy-grid kernels
(morton3D :148-154, morton3D_invert :130-137, packbits :157-169) on the HIP path."""
import os

import numpy as np
import torch

from ngp_hip import ops as _ops

data_type = torch.float32      # the reference exposes ti.f32 here; only used for dtype bookkeeping
torch_type = torch.float32

MAX_SAMPLES = 1024
NEAR_DISTANCE = 0.01
SQRT3 = 1.7320508075688772
SQRT3_MAX_SAMPLES = SQRT3 / 1024
SQRT3_2 = 1.7320508075688772 * 2


def res_in_level_np(level_i, base_res, log_per_level_scale):
    """reference utils.py:19-29"""
    return float(np.ceil(float(base_res) * np.exp(float(level_i) * log_per_level_scale) - 1.0) + 1)


def scale_in_level_np(base_res, max_res, levels):
    """reference utils.py:31-39"""
    return np.log(float(max_res) / float(base_res)) / float(levels - 1)


def align_to(x, y):
    return int((x + y - 1) / y) * y


def morton3D(coords1):
    """[M,3] int32 cell coordinates -> [M] int32 Morton codes."""
    return _ops.morton3d(coords1.contiguous())


def morton3D_invert(indices):
    """[M] int32 Morton codes -> [M,3] int32 cell coordinates."""
    return _ops.morton3d_invert(indices.contiguous())


def packbits(density_grid, density_threshold, density_bitfield):
    """bit i of byte n = density_grid[8n+i] > threshold; writes density_bitfield in place."""
    _ops.packbits(density_grid, density_threshold, density_bitfield)


def depth2img(depth):
    """Turbo-less fallback of reference utils.py:223-228 (cv2 is not a dependency of the hot path)."""
    d = (depth - depth.min()) / max(float(depth.max() - depth.min()), 1e-12)
    g = (d * 255).astype(np.uint8)
    return np.stack([g, g, g], -1)


def save_deployment_model(model, dataset, save_dir):
    """Same dictionary layout as reference utils.py:230-253 so the mobile exporter keeps working."""
    pad = torch.zeros(13, 16)
    rgb_out = torch.cat([model.rgb_net.output_layer.weight.detach().cpu(), pad], dim=0)
    blob = {
        'poses': dataset.poses.cpu().numpy(),
        'model.density_bitfield': model.density_bitfield.cpu().numpy(),
        'model.hash_encoder.params': model.pos_encoder.hash_table.detach().cpu().numpy(),
        'model.per_level_scale': model.pos_encoder.log_b,
        'model.xyz_encoder.params': torch.cat([
            model.xyz_encoder.hidden_layers[0].weight.detach().cpu().reshape(-1),
            model.xyz_encoder.output_layer.weight.detach().cpu().reshape(-1)]).numpy(),
        'model.rgb_net.params': torch.cat([
            model.rgb_net.hidden_layers[0].weight.detach().cpu().reshape(-1),
            rgb_out.reshape(-1)]).numpy(),
    }
    np.save(os.path.join(str(save_dir), 'deployment.npy'), blob)
