"""End to end: one training-mode render through the HIP path (render() -> fused march / hash encode / MFMA MLP / composite)
against the ORACLE pipeline on the same rays, weights, occupancy and jitter noise -- ray_aabb -> march -> hash encode ->
MLPs (numpy, with the fp16 rounding points of torch autocast emulated) -> composite.  BASELINE north_star: indexing /
compaction bit-exact, rendered radiance within 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _h(x):                       # round to fp16, keep computing in fp32 (what an fp16 tensor holds)
    return x.astype(np.float16).astype(np.float32)


def _linear_autocast(x, w):      # torch autocast Linear: fp16 operands, fp32 accumulate, fp16 result
    return _h(_h(x) @ _h(w).T)


def _oracle_render(oracle, model, o, d, bits, noise, T_thr=1e-4):
    scale = float(model.scale)
    hits = oracle.ray_aabb(o, d, scale)
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, noise, model.cascades, scale, 0.0, model.grid_size, 1024)
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    x01 = ((xyzs - (-scale)) / (scale - (-scale))).astype(np.float32)                 # networks.py:144
    enc = oracle.hash_fwd_f32(x01, model.pos_encoder.hash_table.detach().cpu().numpy(), lv)
    W1, W2, W3, W4, W5 = [w.detach().cpu().numpy() for w in model._mlp_weights()]
    h = _linear_autocast(np.maximum(_linear_autocast(enc, W1), 0), W2)                # xyz_encoder 32 -> 64 -> 16
    sigmas = np.exp(h[:, 0].astype(np.float32))                                       # TruncExp on h[:, 0], fp32
    dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    sh = oracle.sh16_fwd(((dn + 1) / 2).astype(np.float32))                           # networks.py:162-163
    x = np.concatenate([sh, h], 1)
    x = np.maximum(_linear_autocast(x, W3), 0)
    x = np.maximum(_linear_autocast(x, W4), 0)
    rgbs = _h(1.0 / (1.0 + np.exp(-_linear_autocast(x, W5))))                         # Sigmoid on an fp16 tensor
    vr, op, dep, rgb, ws = oracle.composite_train_fwd(sigmas, rgbs, deltas, ts, rays_a, T_thr)
    rgb = rgb + 1.0 * (1.0 - op)[:, None]                                             # white background, rendering.py:219-226
    return rays_a, total, rgb, op, dep, int(vr.sum())


@pytest.mark.parametrize("fused", [True, False])
def test_render_matches_oracle_pipeline(oracle, hip_lib, lego_bitfield, fused):
    import os
    from modules.networks import NGP
    from modules.rendering import render
    from ngp_hip import synthetic
    n = 2048
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    o, d = synthetic.lego_rays(n, seed=11)
    torch.manual_seed(99)
    noise = torch.rand(n, device="cuda").cpu().numpy()          # the first draw after the seed is the march jitter
    torch.manual_seed(99)
    os.environ["NGP_FUSED_RENDER"] = "1" if fused else "0"
    m.use_fused_mlp = fused
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            res = render(m, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), exp_step_factor=0.0)
    finally:
        os.environ["NGP_FUSED_RENDER"] = "1"
        m.use_fused_mlp = True
    rays_a, total, rgb, op, dep, vr = _oracle_render(oracle, m, o, d, lego_bitfield, noise)
    # indexing / compaction: bit-exact
    assert int(res["rm_samples"]) == total > 20000
    ra = res["rays_a"].cpu().numpy()
    assert np.array_equal(ra[:, [0, 2]], rays_a[:, [0, 2]])
    if not fused:
        assert np.array_equal(ra, rays_a)        # the operator chain packs in ray order; the fused march in block order
    assert int(res["vr_samples"]) == vr
    # radiance: within 1e-3 (mean), and no outlier beyond a few fp16 ulps of the accumulated colour
    got = res["rgb"].float().detach().cpu().numpy()
    err = np.abs(got - rgb)
    assert err.mean() < 1e-5, err.mean()         # measured 6e-8: the fp16 rounding points coincide, only fp32 summation order differs
    assert err.max() < 1e-4, err.max()           # measured 1e-6
    psnr = -10 * np.log10(np.mean((got - rgb) ** 2))
    print("e2e fused=%s: samples %d, mean|d rgb| %.2e, max %.2e, PSNR(HIP vs oracle) %.1f dB" % (fused, total, err.mean(), err.max(), psnr))
    assert psnr > 100.0, psnr
    np.testing.assert_allclose(res["opacity"].float().detach().cpu().numpy(), op, atol=1e-4)
    np.testing.assert_allclose(res["depth"].float().detach().cpu().numpy(), dep, atol=1e-4)
