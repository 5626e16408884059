"""Camera rays and training-batch sampling on the device.

`get_rays` / `get_ray_directions` keep the signatures of the reference's datasets/ray_utils.py:7-80, so train.py's
`from datasets.ray_utils import get_rays` can be pointed here unchanged; `RayBatcher` is the training-split
`BaseDataset.__getitem__` (datasets/base.py:34-61) + train.py:171-184 as one kernel: two torch.randint draws, then pose /
direction / pixel gathers and the ray transform in a single launch (the reference: three fancy-index gathers, a batched
matmul, an expand and two layout changes)."""
import ctypes

import torch

from . import lib as _lib_mod
from .lib import check
from .ops import _dev, _ptr, _stream


def get_ray_directions(H, W, K, device='cpu', random=False, return_uv=False, flatten=True):
    """Ray directions in camera coordinates [right down front] for every pixel (ray_utils.py:7-48): ((u - cx + 0.5) / fx,
    (v - cy + 0.5) / fy, 1) with u the column and v the row index.  One-time host-side setup, plain torch."""
    v, u = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32), torch.arange(W, device=device, dtype=torch.float32),
                          indexing='ij')
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if random:
        directions = torch.stack([(u - cx + torch.rand_like(u)) / fx, (v - cy + torch.rand_like(v)) / fy, torch.ones_like(u)], -1)
    else:
        directions = torch.stack([(u - cx + 0.5) / fx, (v - cy + 0.5) / fy, torch.ones_like(u)], -1)
    grid = torch.stack([u, v], -1)
    if flatten:
        directions = directions.reshape(-1, 3)
        grid = grid.reshape(-1, 2)
    return (directions, grid) if return_uv else directions


def get_rays(directions, c2w):
    """directions [N,3] (camera frame), c2w [3,4] or [N,3,4] -> rays_o, rays_d [N,3] float32 in world coordinates
    (ray_utils.py:51-80).  Device tensors only."""
    directions = directions.contiguous().float()
    c2w = c2w.float()
    if c2w.shape[-1] == 4 and c2w.shape[-2] == 4:                       # homogeneous 4x4 poses: the reference drops the last row
        c2w = c2w[..., :3, :]
    c2w = c2w.contiguous()
    _dev(directions, torch.float32, "directions"); _dev(c2w, torch.float32, "c2w")
    n = directions.shape[0]
    per_ray = c2w.ndim == 3
    if per_ray and c2w.shape[0] != n:
        raise ValueError("per-ray poses need one [3,4] matrix per direction")
    rays_o = torch.empty(n, 3, device=directions.device, dtype=torch.float32)
    rays_d = torch.empty(n, 3, device=directions.device, dtype=torch.float32)
    check(_lib_mod.load().ngp_get_rays(_ptr(directions), _ptr(c2w), int(per_ray), n, _ptr(rays_o), _ptr(rays_d), _stream()),
          "ngp_get_rays")
    return rays_o, rays_d


class RayBatcher:
    """Device-resident training split: rays [n_img, H*W, C>=3] float32 (rgb first, like BaseDataset.rays), poses [n_img,3,4],
    directions [H*W,3].  `sample()` returns what train.py consumes for one step."""

    def __init__(self, rays, poses, directions, batch_size=8192, ray_sampling_strategy='all_images'):
        if ray_sampling_strategy not in ('all_images', 'same_image'):
            raise ValueError(ray_sampling_strategy)
        self.rays = rays.contiguous().float()
        self.poses = poses[..., :3, :].contiguous().float()
        self.directions = directions.contiguous().float()
        for t, name in ((self.rays, "rays"), (self.poses, "poses"), (self.directions, "directions")):
            _dev(t, torch.float32, name)
        if self.rays.ndim != 3 or self.rays.shape[2] < 3 or self.rays.shape[1] != self.directions.shape[0] \
                or self.rays.shape[0] != self.poses.shape[0]:
            raise ValueError("rays [n_img, H*W, C>=3], poses [n_img,3,4], directions [H*W,3] expected")
        self.batch_size = int(batch_size)
        self.ray_sampling_strategy = ray_sampling_strategy
        self.L = _lib_mod.load()

    def __len__(self):
        return self.poses.shape[0]

    def sample(self, idx=None, generator=None):
        """One training batch.  all_images: an image index per ray; same_image: every ray from image `idx` (drawn like
        train.py:171 when None).  -> {'img_idxs', 'pix_idxs', 'rays_o', 'rays_d', 'rgb'} (device tensors)."""
        dev, n = self.rays.device, self.batch_size
        hw = self.rays.shape[1]
        if self.ray_sampling_strategy == 'all_images':
            img_idxs = torch.randint(0, len(self), (n,), device=dev, generator=generator)            # base.py:40-45
            img_ptr, img0 = _ptr(img_idxs), 0
        else:
            if idx is None:
                idx = int(torch.randint(0, len(self), (1,)).item())                                   # train.py:171
            img_idxs, img_ptr, img0 = idx, _ptr(None), int(idx)
        pix_idxs = torch.randint(0, hw, (n,), device=dev, generator=generator)                       # base.py:51-53
        f32 = dict(device=dev, dtype=torch.float32)
        rays_o, rays_d, rgb = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
        check(self.L.ngp_sample_rays(_ptr(self.poses), _ptr(self.directions), _ptr(self.rays), self.rays.shape[2],
                                     ctypes.c_longlong(hw), img_ptr, ctypes.c_longlong(img0), _ptr(pix_idxs), n, _ptr(rays_o),
                                     _ptr(rays_d), _ptr(rgb), _stream()), "ngp_sample_rays")
        return {'img_idxs': img_idxs, 'pix_idxs': pix_idxs, 'rays_o': rays_o, 'rays_d': rays_d, 'rgb': rgb}
