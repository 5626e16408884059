"""Row f-4 on the GPU: ngp_get_rays / ngp_sample_rays (through ngp_hip.rays) against the oracle -- bit-exact -- and against the
vectors of the reference's get_rays."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_get_rays.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_get_rays_bit_exact_vs_oracle_and_reference_vectors(oracle, hip_lib):
    from ngp_hip.rays import get_rays
    o, d = get_rays(dev(G["directions"]), dev(G["poses"][2]))
    oo, od = oracle.get_rays(G["directions"], G["poses"][2])
    assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(d.cpu().numpy(), od)
    assert np.array_equal(o.cpu().numpy(), G["rays_o_image"]) and np.abs(d.cpu().numpy() - G["rays_d_image"]).max() < 2e-7
    o, d = get_rays(dev(G["directions"][G["pix_idxs"]]), dev(G["poses"][G["img_idxs"]]))
    assert np.array_equal(o.cpu().numpy(), G["rays_o_batch"]) and np.abs(d.cpu().numpy() - G["rays_d_batch"]).max() < 4e-7
    # a full 800 x 800 image and a 65536-ray batch
    rng = np.random.default_rng(1)
    dirs = rng.standard_normal((640000, 3)).astype(np.float32)
    pose = rng.standard_normal((3, 4)).astype(np.float32)
    o, d = get_rays(dev(dirs), dev(pose))
    oo, od = oracle.get_rays(dirs, pose)
    assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(d.cpu().numpy(), od)
    poses = rng.standard_normal((65536, 3, 4)).astype(np.float32)
    o, d = get_rays(dev(dirs[:65536]), dev(poses))
    oo, od = oracle.get_rays(dirs[:65536], poses)
    assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(d.cpu().numpy(), od)
    # 4 x 4 homogeneous poses: the last row is dropped like ray_utils.py:75-77
    p44 = np.concatenate([pose, np.array([[0, 0, 0, 1]], np.float32)], 0)
    o2, d2 = get_rays(dev(dirs[:1000]), dev(p44))
    assert torch.equal(o2, o[:0].new_tensor(np.broadcast_to(pose[:, 3], (1000, 3)).copy())) and np.array_equal(d2.cpu().numpy(),
                                                                                                              oracle.get_rays(dirs[:1000], pose)[1])
    with pytest.raises(Exception):
        get_rays(torch.from_numpy(dirs[:8]), torch.from_numpy(pose))          # host tensors: no CPU path


@pytest.mark.parametrize("strategy", ["all_images", "same_image"])
def test_ray_batcher(oracle, hip_lib, strategy):
    from ngp_hip.rays import RayBatcher
    rng = np.random.default_rng(2)
    n_img, hw = 7, 40 * 30
    rays = rng.random((n_img, hw, 4), dtype=np.float32)
    poses = rng.standard_normal((n_img, 3, 4)).astype(np.float32)
    dirs = rng.standard_normal((hw, 3)).astype(np.float32)
    rb = RayBatcher(dev(rays), dev(poses), dev(dirs), batch_size=4096, ray_sampling_strategy=strategy)
    torch.manual_seed(5)
    s = rb.sample(idx=4 if strategy == "same_image" else None)
    pix = s["pix_idxs"].cpu().numpy()
    img = s["img_idxs"].cpu().numpy() if strategy == "all_images" else 4
    assert pix.min() >= 0 and pix.max() < hw and len(np.unique(pix)) > 500
    o, d, c = oracle.sample_rays(poses, dirs, rays, img, pix)
    assert np.array_equal(s["rays_o"].cpu().numpy(), o) and np.array_equal(s["rays_d"].cpu().numpy(), d)
    assert np.array_equal(s["rgb"].cpu().numpy(), c)
    # same draws as the reference's dataset code under the same seed (torch.randint on the device, base.py:40-53)
    torch.manual_seed(5)
    if strategy == "all_images":
        assert torch.equal(torch.randint(0, n_img, (4096,), device="cuda"), s["img_idxs"])
    assert torch.equal(torch.randint(0, hw, (4096,), device="cuda"), s["pix_idxs"])
