"""The half2 encoder (BASELINE C5, reference modules/hash_encoder_half.py) on the fused path: ngp_hash_fwd_f16_ex / ngp_hash_bwd_f16_ex /
ngp_check_finite_f16 / ngp_adam_all_ex and FusedTrainer(model with half_opt=True) against the operator-path kernels and the
reference-shaped torch loop."""
import copy
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lv():
    from ngp_hip import ops
    return ops.make_levels(2**19, 16, 16, 1024, 2)


def test_fwd_f16_ex_is_the_half_encoder_bit_for_bit(hip_lib):
    from ngp_hip import lib as L, ops
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    lv = _lv()
    torch.manual_seed(0)
    n = 20000
    x = torch.rand(n, 3, device="cuda")
    table = ((torch.rand(lv.total_entries, 2, device="cuda") - 0.5) * 4).half()
    ref = ops.hash_fwd_f16(x, table, lv).view(n, 32).float()                    # operator-path kernel (golden-checked)
    out = torch.empty(n, 32, device="cuda")
    L.check(lib.ngp_hash_fwd_f16_ex(_ptr(x), _ptr(table), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 0, _ptr(out), _stream()), "fwd")
    assert torch.equal(out, ref)
    # pair-major layout + fused normalisation + device-side count
    cap = 24000
    xw = torch.zeros(cap, 3, device="cuda"); xw[:n] = x * 2 - 1
    cnt = torch.tensor([n], device="cuda", dtype=torch.int32)
    outp = torch.zeros(8, cap, 4, device="cuda")
    L.check(lib.ngp_hash_fwd_f16_ex(_ptr(xw), _ptr(table), ctypes.byref(lv), cap, _ptr(cnt), 1, -1.0, 1.0, 1, _ptr(outp), _stream()), "fwd")
    ref2 = ops.hash_fwd_f16(((xw[:n] - (-1.0)) / (1.0 - (-1.0))).contiguous(), table, lv).view(n, 16, 2).float()
    nat = torch.empty(n, 16, 2, device="cuda")
    for p in range(8):
        nat[:, p] = outp[p, :n, 0:2]; nat[:, 15 - p] = outp[p, :n, 2:4]
    assert torch.equal(nat, ref2) and not outp[:, n:].any()


def test_bwd_f16_ex_matches_operator_kernel(hip_lib):
    from ngp_hip import lib as L, ops
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    lv = _lv()
    torch.manual_seed(1)
    n = 30000
    # consecutive samples along rays share cells on the coarse levels: exercise the run merging
    base = torch.rand(n // 30, 1, 3, device="cuda")
    x = (base + torch.linspace(0, 0.05, 30, device="cuda")[None, :, None] * torch.randn(n // 30, 1, 3, device="cuda")).reshape(-1, 3)
    x = x.clamp(0, 1).contiguous()
    dout = torch.randn(n, 32, device="cuda") * 0.01
    dout[::5] = 0
    ref = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(x, dout.half().view(n, 16, 2), lv, ref)
    got = torch.zeros_like(ref)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    L.check(lib.ngp_hash_bwd_f16_ex(_ptr(x), _ptr(dout), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 0, _ptr(got), _ptr(flag), _stream()), "bwd")
    a, b = got.float(), ref.float()
    assert int(flag) == 0
    assert ((a != 0) == (b != 0)).float().mean().item() > 0.999                # same touched entries (up to f16 underflow of merged sums)
    # f16 accumulation order differs (runs are pre-summed in f32): a few f16 ulps of the largest partial sum
    assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    assert ((a - b).norm() / b.norm()).item() < 2e-3
    # non-finite incoming gradient -> flag
    dout[7, 3] = float("inf")
    L.check(lib.ngp_hash_bwd_f16_ex(_ptr(x), _ptr(dout), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 0, _ptr(got), _ptr(flag), _stream()), "bwd")
    assert int(flag) == 1


def test_check_finite_f16(hip_lib):
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    g = (torch.randn(1 << 20, device="cuda") * 100).half()
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    L.check(lib.ngp_check_finite_f16(_ptr(g), g.numel(), _ptr(flag), _stream()), "check")
    assert int(flag) == 0
    for bad in (float("inf"), float("-inf"), float("nan")):
        g2 = g.clone(); g2[123457] = bad
        flag.zero_()
        L.check(lib.ngp_check_finite_f16(_ptr(g2), g2.numel(), _ptr(flag), _stream()), "check")
        assert int(flag) == 1
    g3 = g.clone(); g3[5] = 65504.0                                               # the largest finite f16 is fine
    flag.zero_()
    L.check(lib.ngp_check_finite_f16(_ptr(g3), g3.numel(), _ptr(flag), _stream()), "check")
    assert int(flag) == 0


def test_adam_all_ex_f16_grad_and_copy(hip_lib):
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    n = 1 << 18
    torch.manual_seed(2)
    mk = lambda k: torch.randn(k, device="cuda")
    p, g16, m, v = mk(n), (mk(n) * 64).half(), mk(n).abs() * 0.1, mk(n).abs() * 0.01
    g16[: n // 4] = 0; m[: n // 4] = 0; v[: n // 4] = 0
    wp, wg, wm, wv = mk(9408) * 0.2, mk(9408) * 64, mk(9408) * 0.1, mk(9408).abs() * 0.01
    sf = torch.zeros(8, device="cuda"); si = torch.zeros(8, device="cuda", dtype=torch.int32)
    sf[0] = 64.0
    L.check(lib.ngp_train_prologue(_ptr(sf), _ptr(si), 1e-2, 1e-2 / 30, 100, 0.9, 0.999, 2.0, 0.5, 2000, _stream()), "prologue")
    A = [t.clone() for t in (p, g16, m, v, wp, wg, wm, wv)]
    B = [t.clone() for t in (p, g16.float(), m, v, wp, wg, wm, wv)]
    copy16 = p.half()
    nh = lib.ngp_mlp_wpack_halfs()
    wk_a, wk_b = torch.zeros(nh, device="cuda", dtype=torch.float16), torch.zeros(nh, device="cuda", dtype=torch.float16)
    L.check(lib.ngp_adam_all_ex(_ptr(A[0]), _ptr(A[1]), 1, _ptr(A[2]), _ptr(A[3]), n, _ptr(copy16), 2, _ptr(A[4]), _ptr(A[5]), _ptr(A[6]),
                                _ptr(A[7]), _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, 1, _ptr(wk_a), _stream()), "adam_all_ex")
    L.check(lib.ngp_adam_all(_ptr(B[0]), _ptr(B[1]), _ptr(B[2]), _ptr(B[3]), n, _ptr(None), _ptr(B[4]), _ptr(B[5]), _ptr(B[6]), _ptr(B[7]),
                             _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, 1, _ptr(wk_b), _stream()), "adam_all")
    torch.cuda.synchronize()
    for k in (0, 2, 3, 4, 6, 7):
        assert torch.equal(A[k], B[k])
    assert not A[1].any() and not B[1].any() and torch.equal(wk_a.view(torch.int16), wk_b.view(torch.int16))
    assert torch.equal(copy16.view(torch.int16), A[0].half().view(torch.int16))


def test_trainer_half_matches_torch_loop(hip_lib, lego_bitfield):
    """FusedTrainer on the half2-encoder model vs train.py's loop shape over the operator-path half encoder."""
    from modules.networks import NGP
    from modules.rendering import render
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    n = 4096
    torch.manual_seed(0)
    m_a = NGP(scale=0.5, max_res=1024, half_opt=True).cuda()
    m_a.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m_a.pos_encoder.hash_table.uniform_(0.0, 0.2)         # (the reference's 1e-4 init renders an almost uniform fog)
    m_b = copy.deepcopy(m_a)
    o, d = synthetic.lego_rays(n, seed=9)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(n, 3, device="cuda")
    steps, T, scale0 = 6, 50, 2.0**7
    opt = torch.optim.Adam(m_a.parameters(), 1e-2, eps=1e-15)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T, 1e-2 / 30)
    scaler = torch.amp.GradScaler("cuda", init_scale=scale0)
    losses_a = []
    for i in range(steps):
        torch.manual_seed(100 + i)
        with torch.autocast("cuda", dtype=torch.float16):
            res = render(m_a, o, d, exp_step_factor=0.0)
            loss = F.mse_loss(res["rgb"], target)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); sched.step()
        losses_a.append(loss.item())
    tr = FusedTrainer(m_b, lr=1e-2, max_steps=T, init_scale=scale0)
    assert tr.half and tr.table_grad.dtype == torch.float16
    losses_b = []
    for i in range(steps):
        torch.manual_seed(100 + i)
        tr.step(o, d, target, noise=torch.rand(o.shape[0], device="cuda"))            # the jitter render() drew for this seed
        losses_b.append(tr.last_loss())
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses_b, losses_a, rtol=3e-2, atol=3e-3)
    assert losses_b[-1] < losses_b[0]
    assert tr.counters()["skipped"] == 0 and scaler.get_scale() == scale0
    ta, tb = m_a.pos_encoder.hash_table.detach(), m_b.pos_encoder.hash_table.detach()
    rel = ((ta - tb).norm() / ta.norm()).item()
    assert rel < 3e-2, rel
    # the f16 copy the forward gathers from is the parameter, cast
    assert torch.equal(tr.table_f16.view(torch.int16), tb.reshape(-1).half().view(torch.int16))
    # an fp16 overflow anywhere in the backward is caught, the step skipped, the scale backed off, nothing poisoned
    tr2 = FusedTrainer(copy.deepcopy(m_b), init_scale=2.0**32)
    before = tr2.table.clone()
    tr2.step(o, d, target)
    assert tr2.counters()["skipped"] == 1 and tr2.loss_scale() == 2.0**31 and torch.equal(before, tr2.table)
    assert not tr2.table_grad.any() and torch.isfinite(tr2.table_m).all()
    # the occupancy update runs on the half encoder's fused kernels too
    tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=True)
    torch.cuda.synchronize()
    assert torch.isfinite(m_b.density_grid).all() and (m_b.density_grid > 0).any()


def test_render_fused_half_matches_operator_path(hip_lib, lego_bitfield):
    """`render(NGP(half_opt=True))` -- what the reference's unchanged `train.py --half_opt` calls (README.md:39-42: its fastest
    mode) -- through the fused render (the half2 encoder's kernels inside one autograd node, round 4) against the operator chain
    (modules/hash_encoder_half.py): same samples, radiance within fp16 tolerance, table gradient with the same support."""
    import os
    import torch.nn.functional as F
    from conftest import ray_order
    from modules.networks import NGP
    from modules.rendering import render
    from ngp_hip import synthetic
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024, half_opt=True).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.uniform_(-0.15, 0.15)          # (the reference's 1e-4 init gives no signal to compare)
    o, d = synthetic.lego_rays(4096, seed=3)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(4096, 3, device="cuda")

    def run(fused):
        for p in m.parameters():
            p.grad = None
        torch.manual_seed(123)
        os.environ["NGP_FUSED_RENDER"] = "1" if fused else "0"
        try:
            with torch.autocast("cuda", dtype=torch.float16):
                assert m.fused_train_ok(o) == fused
                res = render(m, o, d, exp_step_factor=0.0)
                loss = F.mse_loss(res["rgb"], target)
            (loss * 256.0).backward()
        finally:
            os.environ["NGP_FUSED_RENDER"] = "1"
        return res, [p.grad.clone().float() for p in [m.pos_encoder.hash_table, *m._mlp_weights()]]

    r_f, g_f = run(True)
    r_o, g_o = run(False)
    assert torch.equal(r_f["rays_a"][:, [0, 2]], r_o["rays_a"][:, [0, 2]]) and int(r_f["rm_samples"]) == int(r_o["rm_samples"]) > 0
    pf, po = [torch.from_numpy(ray_order(r["rays_a"])).cuda() for r in (r_f, r_o)]
    assert torch.equal(r_f["ts"][pf], r_o["ts"][po])
    torch.testing.assert_close(r_f["rgb"], r_o["rgb"], rtol=0, atol=4e-3)
    torch.testing.assert_close(r_f["opacity"], r_o["opacity"], rtol=0, atol=4e-3)
    assert g_f[0].shape == m.pos_encoder.hash_table.shape and torch.isfinite(g_f[0]).all()
    nz_f, nz_o = g_f[0] != 0, g_o[0] != 0
    assert (nz_f != nz_o).float().mean().item() < 2e-3
    for k in range(6):
        rel = ((g_f[k] - g_o[k]).norm() / g_o[k].norm().clamp_min(1e-30)).item()
        assert rel < 3e-2, (k, rel)
