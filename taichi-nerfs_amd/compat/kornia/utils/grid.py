import torch


def create_meshgrid3d(depth, height, width, normalized_coordinates=True, device="cpu", dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    zs = torch.linspace(0, depth - 1, depth, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / max(width - 1, 1) - 0.5) * 2
        ys = (ys / max(height - 1, 1) - 0.5) * 2
        zs = (zs / max(depth - 1, 1) - 0.5) * 2
    base = torch.stack(torch.meshgrid([zs, xs, ys], indexing="ij"), dim=-1)   # D x W x H x 3
    return base.permute(0, 2, 1, 3).unsqueeze(0)                              # 1 x D x H x W x 3
