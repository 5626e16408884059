"""Row f-4 (camera rays): the oracle's get_rays / sample_rays against vectors produced by the REFERENCE's own
datasets/ray_utils.py (oracle/gen_golden_rays.py), and the host-side get_ray_directions mirror.  CPU only."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_get_rays.npz"))


def _ulps(a, b):
    return np.abs(a.astype(np.float64) - b) / np.maximum(np.spacing(np.abs(b).astype(np.float32)), 1e-45)


def test_oracle_get_rays_matches_reference(oracle):
    # one pose per image (evaluation) and one pose per ray (training batch)
    o, d = oracle.get_rays(G["directions"], G["poses"][2])
    assert np.array_equal(o, G["rays_o_image"])
    # the reference's matmul leaves the summation order to the BLAS: agreement to the last bit or two, not bit equality
    assert _ulps(d, G["rays_d_image"]).max() <= 2 or np.abs(d - G["rays_d_image"]).max() < 2e-7
    o, d = oracle.get_rays(G["directions"][G["pix_idxs"]], G["poses"][G["img_idxs"]])
    assert np.array_equal(o, G["rays_o_batch"])
    assert np.abs(d - G["rays_d_batch"]).max() < 2e-7 * max(1.0, np.abs(G["rays_d_batch"]).max())


def test_oracle_sample_rays_is_gather_plus_get_rays(oracle):
    rng = np.random.default_rng(0)
    hw = G["directions"].shape[0]
    rays = rng.random((G["poses"].shape[0], hw, 4), dtype=np.float32)                  # rgba images
    o, d, c = oracle.sample_rays(G["poses"], G["directions"], rays, G["img_idxs"], G["pix_idxs"])
    o2, d2 = oracle.get_rays(G["directions"][G["pix_idxs"]], G["poses"][G["img_idxs"]])
    assert np.array_equal(o, o2) and np.array_equal(d, d2)
    assert np.array_equal(c, rays[G["img_idxs"], G["pix_idxs"]][:, :3])
    o, d, c = oracle.sample_rays(G["poses"], G["directions"], rays, 3, G["pix_idxs"])  # 'same_image'
    o2, d2 = oracle.get_rays(G["directions"][G["pix_idxs"]], G["poses"][3])
    assert np.array_equal(o, o2) and np.array_equal(d, d2) and np.array_equal(c, rays[3, G["pix_idxs"], :3])


def test_get_ray_directions_mirror_matches_reference():
    from ngp_hip.rays import get_ray_directions
    K = torch.from_numpy(G["K"])
    H, W = int(G["H"]), int(G["W"])
    assert np.array_equal(get_ray_directions(H, W, K).numpy(), G["directions"])
    d, uv = get_ray_directions(H, W, K, return_uv=True, flatten=False)
    assert np.array_equal(d.numpy(), G["directions_hw"]) and np.array_equal(uv.numpy(), G["uv"])
    r = get_ray_directions(H, W, K, random=True)
    assert r.shape == (H * W, 3) and float((r - torch.from_numpy(G["directions"])).abs().max()) <= 0.5 / 13.5 + 1e-6


def test_ray_helpers_have_no_cpu_path():
    """The product path fails loudly on host tensors instead of falling back."""
    import pytest
    from ngp_hip.rays import RayBatcher, get_rays
    d = torch.from_numpy(G["directions"])
    with pytest.raises(Exception):
        get_rays(d, torch.from_numpy(G["poses"][0]))
    with pytest.raises(Exception):
        RayBatcher(torch.rand(5, d.shape[0], 3), torch.from_numpy(G["poses"]), d)
