"""Golden vectors for row f-4 (camera rays): runs the REFERENCE's own datasets/ray_utils.py (get_ray_directions, get_rays) on
the CPU in this container and stores inputs + outputs under tests/golden/ref_get_rays.npz.  The reference file is loaded by
path (never copied); its `kornia.create_meshgrid` import is satisfied by taichi-nerfs_amd/compat.
    python oracle/gen_golden_rays.py            (needs /root/reference; test infrastructure only)"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NGP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    sys.path.insert(0, os.path.join(ROOT, "taichi-nerfs_amd", "compat"))
    spec = importlib.util.spec_from_file_location("ref_ray_utils", os.path.join(REF, "datasets", "ray_utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    rng = np.random.default_rng(77)
    H, W = 10, 12
    K = np.array([[13.5, 0, 6.2], [0, 14.25, 4.9], [0, 0, 1]], np.float32)
    directions = ru.get_ray_directions(H, W, torch.from_numpy(K))                        # [H*W, 3]
    dirs_hw, uv = ru.get_ray_directions(H, W, torch.from_numpy(K), return_uv=True, flatten=False)
    # random rigid poses
    n_img = 5
    q = rng.standard_normal((n_img, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    Rm = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    poses = np.concatenate([Rm, rng.standard_normal((n_img, 3, 1)) * 1.5], -1).astype(np.float32)
    # (a) one pose for a whole image (train.py:253)
    o1, d1 = ru.get_rays(directions, torch.from_numpy(poses[2]))
    # (b) per-ray poses (train.py:184 with datasets/base.py:34-61 gathers)
    n = 64
    img_idxs = rng.integers(0, n_img, n)
    pix_idxs = rng.integers(0, H * W, n)
    o2, d2 = ru.get_rays(directions[pix_idxs], torch.from_numpy(poses)[img_idxs])
    np.savez_compressed(os.path.join(OUT, "ref_get_rays.npz"), H=H, W=W, K=K, directions=directions.numpy(),
                        directions_hw=dirs_hw.numpy(), uv=uv.numpy(), poses=poses, rays_o_image=o1.numpy().copy(),
                        rays_d_image=d1.numpy(), img_idxs=img_idxs.astype(np.int64), pix_idxs=pix_idxs.astype(np.int64),
                        rays_o_batch=o2.numpy().copy(), rays_d_batch=d2.numpy())
    print("get_rays ok", directions.shape, o1.shape, o2.shape)


if __name__ == "__main__":
    main()
