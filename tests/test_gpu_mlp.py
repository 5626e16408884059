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


def _to_pairs(enc_nat, n_max):
    """[n,32] natural (level-major) -> pair-major planes [8, n_max, 4] (include/ngp_hip.h, enc_pairs)."""
    n = enc_nat.shape[0]
    out = torch.zeros(8, n_max, 4, device=enc_nat.device, dtype=enc_nat.dtype)
    for p in range(8):
        out[p, :n, 0:2] = enc_nat[:, 2 * p:2 * p + 2]
        out[p, :n, 2:4] = enc_nat[:, 2 * (15 - p):2 * (15 - p) + 2]
    return out


def test_pair_major_layout_equals_natural_layout(hip_lib):
    """The fused path's pair-major encoding planes are a pure re-indexing: hash fwd bit-exact, MLP fwd/bwd equal up to the
    fp32 summation order inside one K=32 MFMA step, hash bwd equal up to atomic order."""
    import ctypes
    from ngp_hip import lib as L_, ops
    from ngp_hip.ops import _ptr, _stream, check
    L = L_.load()
    m = _model()
    n = 20000
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    x = torch.rand(n, 3, device="cuda")
    table = torch.rand(lv.total_entries * 2, device="cuda")
    enc_nat = ops.hash_fwd_f32(x, table, lv)
    enc_p = torch.zeros(8, n, 4, device="cuda")
    check(L.ngp_hash_fwd_f32_ex(_ptr(x), _ptr(table), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 1, _ptr(enc_p), _stream()), "fwd")
    assert torch.equal(enc_p, _to_pairs(enc_nat, n))
    dirs = torch.randn(n, 3, device="cuda")
    ws = [w.detach().contiguous() for w in m._mlp_weights()]
    wp_nat = ops.mlp_pack(ws)
    wp_pair = torch.empty_like(wp_nat)
    check(L.ngp_mlp_pack(*[_ptr(w) for w in ws], 1, _ptr(wp_pair), _stream()), "pack")
    s_nat, c_nat = ops.mlp_fwd(enc_nat, dirs, wp_nat)
    s_p = torch.empty(n, device="cuda"); c_p = torch.empty(n, 3, device="cuda", dtype=torch.float16)
    check(L.ngp_mlp_fwd_ex(_ptr(enc_p), _ptr(dirs), _ptr(wp_pair), n, _ptr(None), 1, _ptr(s_p), _ptr(c_p), _stream()), "mlp fwd")
    torch.testing.assert_close(s_p, s_nat, rtol=2e-3, atol=1e-4)
    torch.testing.assert_close(c_p.float(), c_nat.float(), rtol=0, atol=2e-3)
    g_s = torch.randn(n, device="cuda") * 8
    g_c = (torch.randn(n, 3, device="cuda") * 8).half()
    de_nat, dw_nat = ops.mlp_bwd(enc_nat, dirs, wp_nat, g_s, g_c)
    de_p = torch.zeros(8, n, 4, device="cuda"); dw_p = torch.zeros_like(dw_nat)
    check(L.ngp_mlp_bwd_ex(_ptr(enc_p), _ptr(dirs), _ptr(wp_pair), _ptr(g_s), _ptr(g_c), n, _ptr(None), 1, _ptr(de_p), _ptr(dw_p),
                           _ptr(None), _stream()), "mlp bwd")
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(de_p, _to_pairs(de_nat, n)) < 3e-3
    assert rel(dw_p, dw_nat) < 3e-3
    for k, (lo, hi) in enumerate(zip([0, 2048, 3072, 5120, 9216], [2048, 3072, 5120, 9216, 9408])):
        assert rel(dw_p[lo:hi], dw_nat[lo:hi]) < 5e-3, k                 # W1's column permutation included
    gt_nat = torch.zeros_like(table); gt_p = torch.zeros_like(table)
    ops.hash_bwd_f32(x, de_nat, lv, gt_nat)
    check(L.ngp_hash_bwd_f32_ex(_ptr(x), _ptr(_to_pairs(de_nat, n)), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 1, _ptr(gt_p),
                                _ptr(None), _stream()), "hash bwd")
    torch.testing.assert_close(gt_p, gt_nat, rtol=1e-4, atol=1e-5 * gt_nat.abs().max().item())


@pytest.mark.parametrize("form", ["lds"])       # ("reg": round 4's register-resident form, only in -DNGP_MLP_BWD_REG builds now)
@pytest.mark.parametrize("n", [1, 47, 20000])
def test_backward_forms_and_slab_reduction(hip_lib, monkeypatch, form, n):
    """The backward kernel (the LDS-image form; a -DNGP_MLP_BWD_REG build also answers NGP_EXPERIMENT mlp_bwd=reg) and both
    ways the weight gradients leave it -- float atomics on dW, or per-block slabs + ngp_mlp_dw_reduce (what the trainer
    uses) -- give the same d_enc bit for bit and the same dW up to the summation order; a live list in reverse order too."""
    import ctypes
    from ngp_hip import lib, ops
    from ngp_hip.ops import _ptr, _stream
    L = lib.load()
    m = _model()
    enc, dirs = _inputs(n, seed=5)
    g = torch.Generator(device="cpu").manual_seed(11)
    g_sig = (torch.randn(n, generator=g) * 64).cuda()
    g_rgb = (torch.randn(n, 3, generator=g) * 64).half().cuda()
    wpack = ops.mlp_pack(m._mlp_weights())
    monkeypatch.setenv("NGP_EXPERIMENT", "mlp_bwd=lds")
    de_ref, dw_ref = ops.mlp_bwd(enc, dirs, wpack, g_sig, g_rgb)
    monkeypatch.setenv("NGP_EXPERIMENT", "mlp_bwd=%s" % form)
    de, dw = ops.mlp_bwd(enc, dirs, wpack, g_sig, g_rgb)
    assert torch.equal(de, de_ref)
    scale = dw_ref.abs().max().item() + 1e-30
    assert (dw - dw_ref).abs().max().item() <= 2e-6 * scale
    # slabs + reduction, over a reversed live list
    idx = torch.arange(n - 1, -1, -1, device="cuda", dtype=torch.int32)
    n_dev = torch.tensor([n], device="cuda", dtype=torch.int32)
    parts = torch.full((L.ngp_mlp_dw_parts_max() * 9408,), float("nan"), device="cuda")
    d_enc = torch.empty(n, 32, device="cuda")
    found = torch.zeros(1, device="cuda", dtype=torch.int32)
    n_parts = L.ngp_mlp_bwd_live_parts(_ptr(enc), _ptr(dirs), _ptr(wpack), _ptr(g_sig), _ptr(g_rgb), n, _ptr(n_dev), _ptr(idx), 0,
                                       _ptr(d_enc), _ptr(parts), _ptr(found), _stream())
    assert 1 <= n_parts <= L.ngp_mlp_dw_parts_max()
    dw2 = torch.zeros(9408, device="cuda")
    lib.check(L.ngp_mlp_dw_reduce(_ptr(parts), n_parts, _ptr(dw2), _stream()), "ngp_mlp_dw_reduce")
    assert int(found[0]) == 0
    assert torch.equal(d_enc.flip(0), de_ref)                       # position j of the list = sample n-1-j
    assert torch.isfinite(dw2).all() and (dw2 - dw_ref).abs().max().item() <= 2e-6 * scale
