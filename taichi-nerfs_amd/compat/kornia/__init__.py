"""Stand-in for the two kornia helpers the reference's driver uses (datasets/ray_utils.py:4, modules/networks.py:6)."""
import torch

from . import utils  # noqa: F401


def create_meshgrid(height, width, normalized_coordinates=True, device="cpu", dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / max(width - 1, 1) - 0.5) * 2
        ys = (ys / max(height - 1, 1) - 0.5) * 2
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")
    return torch.stack([gx, gy], dim=-1).permute(1, 0, 2).unsqueeze(0)      # 1 x H x W x 2 (x, y)
