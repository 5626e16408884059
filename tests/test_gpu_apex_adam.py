"""compat/apex FusedAdam (what the reference's unchanged train.py:143-149 picks up) vs torch.optim.Adam(eps=1e-15) under
torch.cuda.amp.GradScaler: same parameters after 50 steps including one overflow step that both must skip."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd", "compat"))


def test_fused_adam_matches_torch_adam_under_gradscaler(hip_lib):
    from apex.optimizers import FusedAdam
    torch.manual_seed(3)
    shapes = [(4096, 2), (64, 32), (16, 64), (3, 64)]
    ref = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, 1e-2, eps=1e-15)
    o_mine = FusedAdam(mine, lr=1e-2, eps=1e-15)
    s_ref, s_mine = torch.cuda.amp.GradScaler(2.0**10, growth_interval=7), torch.cuda.amp.GradScaler(2.0**10, growth_interval=7)
    sch_ref = torch.optim.lr_scheduler.CosineAnnealingLR(o_ref, 50, 1e-2 / 30)
    sch_mine = torch.optim.lr_scheduler.CosineAnnealingLR(o_mine, 50, 1e-2 / 30)
    g = torch.Generator(device="cuda").manual_seed(5)
    for step in range(50):
        tgt = [torch.randn(*s, device="cuda", generator=g) for s in shapes]
        blow = float("inf") if step == 17 else 1.0
        before = [[p.detach().clone() for p in params] for params in (ref, mine)] if step == 17 else None
        for params, opt, scaler, sch in ((ref, o_ref, s_ref, sch_ref), (mine, o_mine, s_mine, sch_mine)):
            loss = sum(((p - t) ** 2).mean() for p, t in zip(params, tgt)) * blow
            opt.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            sch.step()
        if step == 17:                 # the overflow step is skipped on the device: neither optimizer moves a parameter
            for params, snap in zip((ref, mine), before):
                assert all(torch.equal(a.detach(), b) for a, b in zip(params, snap))
        if step == 30:                 # the device-side step count travels with state_dict(): a resumed optimizer goes on
            sd = o_mine.state_dict()   # with the same bias corrections
            assert "ngp_group_state" in sd and int(sd["ngp_group_state"][0][1][1]) == 30
            # ... also from a checkpoint that went through torch.save / torch.load(map_location='cpu') (the device-side
            # counters used to stay HOST tensors after this, and the kernels were handed their host pointers)
            import io
            buf = io.BytesIO()
            torch.save(sd, buf)
            buf.seek(0)
            sd_cpu = torch.load(buf, map_location="cpu")
            assert not sd_cpu["ngp_group_state"][0][0].is_cuda
            o_mine.load_state_dict(sd_cpu)
            assert int(o_mine._si[0][1]) == 30
            assert all(t.is_cuda and t.device == mine[0].device for t in o_mine._sf + o_mine._si)
            assert o_mine._sf[0].dtype == torch.float32 and o_mine._si[0].dtype == torch.int32
            bad = dict(sd_cpu, ngp_group_state=sd_cpu["ngp_group_state"] * 2)
            with pytest.raises(ValueError):
                o_mine.load_state_dict(bad)
            o_mine.load_state_dict(sd_cpu)
    assert s_ref.get_scale() == s_mine.get_scale()
    for a, b in zip(ref, mine):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    assert int(o_mine._si[0][5]) == 1                                   # exactly one skipped step (the overflow)
    assert int(o_mine._si[0][1]) == 49


def test_fused_adam_takes_the_scaler_and_runs_its_own_inf_check(hip_lib):
    """Round 5: GradScaler.step() hands itself to FusedAdam.step(grad_scaler=...) (apex's signature) and then skips its own
    `_check_inf_per_device`; the optimizer's read-only check (ngp_check_finite_multi) records the flag where GradScaler.update()
    reads it.  Also the other order a training loop may use: scaler.unscale_(optimizer) first (gradient clipping), then step."""
    import warnings
    from apex.optimizers import FusedAdam
    torch.manual_seed(4)
    ref = [torch.nn.Parameter(torch.randn(512, 4, device="cuda")), torch.nn.Parameter(torch.randn(64, device="cuda"))]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref, o_mine = torch.optim.Adam(ref, 1e-2, eps=1e-15), FusedAdam(mine, lr=1e-2, eps=1e-15)
    s_ref, s_mine = torch.amp.GradScaler("cuda", init_scale=2.0**12, growth_interval=5), torch.amp.GradScaler("cuda", init_scale=2.0**12, growth_interval=5)
    calls = []
    orig = s_mine._check_inf_per_device
    s_mine._check_inf_per_device = lambda opt: (calls.append(1), orig(opt))[1]
    g = torch.Generator(device="cuda").manual_seed(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)            # torch's notice that the keyword will go away one day
        for step in range(24):
            tgt = [torch.randn_like(p, generator=None) if False else torch.randn(p.shape, device="cuda", generator=g) for p in ref]
            blow = float("nan") if step in (6, 15) else 1.0
            for params, opt, scaler in ((ref, o_ref, s_ref), (mine, o_mine, s_mine)):
                loss = sum(((p - t) ** 2).mean() for p, t in zip(params, tgt)) * blow
                opt.zero_grad()
                scaler.scale(loss).backward()
                if step % 2:                                       # every other step: unscale first, as a clipping loop would
                    scaler.unscale_(opt)
                scaler.step(opt)
                scaler.update()
    assert len(calls) == 0                                         # torch's own check pass never ran for the compat optimizer
    assert s_ref.get_scale() == s_mine.get_scale()
    for a, b in zip(ref, mine):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    assert int(o_mine._si[0][5]) == 2 and int(o_mine._si[0][1]) == 22


def test_fused_adam_without_scaler_and_rejects_unsupported(hip_lib):
    from apex.optimizers import FusedAdam
    p = torch.nn.Parameter(torch.randn(256, device="cuda"))
    q = torch.nn.Parameter(p.detach().clone())
    o1, o2 = FusedAdam([p], lr=1e-3, eps=1e-15), torch.optim.Adam([q], 1e-3, eps=1e-15)
    for _ in range(5):
        for x, o in ((p, o1), (q, o2)):
            o.zero_grad(); (x ** 2).sum().backward(); o.step()
    torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-7)
    with pytest.raises(NotImplementedError):
        FusedAdam([p], weight_decay=0.1)
    odd = torch.nn.Parameter(torch.randn(7, device="cuda"))
    o3 = FusedAdam([odd])
    odd.grad = torch.ones_like(odd)
    with pytest.raises(NotImplementedError):
        o3.step()
