"""Fused MFMA MLP (ngp_mlp_fwd / ngp_mlp_bwd) vs torch: (a) the reference's own formulation -- nn.Linear layers
under torch.autocast(fp16) (modules/networks.py:136-166, train.py:177) -- and (b) a plain fp32 restatement.
Tolerances are fp16 ones: operands and layer outputs are rounded to fp16 on both sides, only the summation order
and the exp/sigmoid implementations differ."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed=0):
    from modules.networks import NGP
    torch.manual_seed(seed)
    m = NGP(scale=0.5, max_res=1024).cuda()
    with torch.no_grad():                         # make the small heads matter (xavier init leaves rgb ~ 0.5 everywhere)
        for w in m._mlp_weights():
            w.mul_(2.0)
    return m


def _inputs(n, seed=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    enc = (torch.rand(n, 32, generator=g) * 2 - 0.5).cuda()
    dirs = torch.randn(n, 3, generator=g).cuda()
    return enc, dirs


def _torch_path(m, enc, dirs):
    from modules.networks import TruncExp
    h = m.xyz_encoder(enc)
    sigmas = TruncExp.apply(h[:, 0])
    d = dirs / torch.norm(dirs, dim=1, keepdim=True)
    sh = m.dir_encoder((d + 1) / 2)
    return sigmas, m.rgb_net(torch.cat([sh, h], 1))


@pytest.mark.parametrize("n", [1, 31, 32, 4097, 70000])
def test_fused_forward_matches_autocast_and_fp32(hip_lib, n):
    from modules.networks import _FusedShade
    m = _model()
    enc, dirs = _inputs(n)
    with torch.no_grad():
        s_f, c_f = _FusedShade.apply(enc, dirs, *m._mlp_weights())
        with torch.autocast("cuda", dtype=torch.float16):
            s_a, c_a = _torch_path(m, enc, dirs)
        s_32, c_32 = _torch_path(m, enc, dirs)
    assert s_f.dtype == torch.float32 and c_f.dtype == torch.float16 and c_f.shape == (n, 3)
    torch.testing.assert_close(s_f, s_a.float(), rtol=2e-2, atol=1e-3)
    torch.testing.assert_close(c_f.float(), c_a.float(), rtol=1e-2, atol=4e-3)
    torch.testing.assert_close(s_f, s_32, rtol=5e-2, atol=1e-2)
    torch.testing.assert_close(c_f.float(), c_32, rtol=2e-2, atol=1e-2)


def test_fused_density_only(hip_lib):
    from ngp_hip import ops
    m = _model()
    enc, dirs = _inputs(5000)
    wpack = ops.mlp_pack(m._mlp_weights())
    s_full, _ = ops.mlp_fwd(enc, dirs, wpack)
    assert torch.equal(ops.mlp_density(enc, wpack), s_full)


@pytest.mark.parametrize("n", [33, 20000])
def test_fused_backward_matches_autocast_and_fp32(hip_lib, n):
    from modules.networks import _FusedShade
    m = _model()
    enc, dirs = _inputs(n, seed=3)
    g = torch.Generator(device="cpu").manual_seed(7)
    g_sig = (torch.randn(n, generator=g) * 64).cuda()                     # loss-scaled magnitudes, like GradScaler's
    g_rgb = (torch.randn(n, 3, generator=g) * 64).cuda()
    ws = list(m._mlp_weights())

    def run(mode):
        e = enc.clone().requires_grad_(True)
        for w in ws:
            w.grad = None
        if mode == "fused":
            s, c = _FusedShade.apply(e, dirs, *ws)
        elif mode == "autocast":
            with torch.autocast("cuda", dtype=torch.float16):
                s, c = _torch_path(m, e, dirs)
        else:
            s, c = _torch_path(m, e, dirs)
        torch.autograd.backward([s, c], [g_sig.to(s.dtype), g_rgb.to(c.dtype)])
        return e.grad.clone(), [w.grad.clone().float() for w in ws]

    de_f, dw_f = run("fused")
    de_a, dw_a = run("autocast")
    de_32, dw_32 = run("fp32")

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()

    # against fp32 ground truth the fused path must be at least as close as torch's own autocast path (+ slack)
    assert rel(de_f, de_32) < max(2.0 * rel(de_a, de_32), 2e-3), (rel(de_f, de_32), rel(de_a, de_32))
    for k in range(5):
        assert rel(dw_f[k], dw_32[k]) < max(2.0 * rel(dw_a[k], dw_32[k]), 2e-3), (k, rel(dw_f[k], dw_32[k]), rel(dw_a[k], dw_32[k]))
    assert rel(de_f, de_a) < 2e-2
    # element-wise on the small weight gradients (asymmetric weights catch any row/column or fragment swap)
    # the worst element may not be worse than 3x torch-autocast's own worst element (fp16 rounding of dZ dominates both)
    for k in range(5):
        err_f = (dw_f[k] - dw_32[k]).abs().max().item()
        err_a = (dw_a[k] - dw_32[k]).abs().max().item()
        assert err_f <= 3.0 * err_a + 1e-3 * dw_32[k].abs().max().item(), (k, err_f, err_a)


def test_ngp_forward_uses_fused_path_under_autocast(hip_lib):
    m = _model()
    x = (torch.rand(3000, 3, device="cuda") - 0.5) * 0.98
    d = torch.randn(3000, 3, device="cuda")
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            s1, c1 = m(x, d)
            m.use_fused_mlp = False
            s2, c2 = m(x, d)
            m.use_fused_mlp = True
            dens = m.density(x)
    torch.testing.assert_close(s1, s2.float(), rtol=2e-2, atol=1e-3)
    torch.testing.assert_close(c1.float(), c2.float(), rtol=1e-2, atol=4e-3)
    torch.testing.assert_close(dens, s1, rtol=0, atol=0)
