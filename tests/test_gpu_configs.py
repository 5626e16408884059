"""The other BASELINE configs as parity cases: C3 (360_v2 Garden shape: scale 16, 6 cascades, max_res 4096, exponential
stepping, black background, distortion loss) and C5 (half2 hash encoder) run end to end through the drop-in modules."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _garden(n=4096):
    from modules.networks import NGP
    from ngp_hip import synthetic
    torch.manual_seed(0)
    m = NGP(scale=16.0, max_res=4096).cuda()
    assert m.cascades == 6 and m.pos_encoder.total_param_size == 12599920          # SURVEY section 8 header
    m.density_bitfield.copy_(torch.from_numpy(synthetic.ball_slab_bitfield(6, 16.0, seed=7)).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    o, d = synthetic.garden_rays(n, seed=5)
    return m, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.rand(n, 3, device="cuda")


def _step(m, o, d, target, fused, w_dist):
    import os
    from modules.distortion import distortion_loss
    from modules.rendering import render
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(7)
    os.environ["NGP_FUSED_RENDER"] = "1" if fused else "0"
    m.use_fused_mlp = fused
    with torch.autocast("cuda", dtype=torch.float16):
        res = render(m, o, d, exp_step_factor=1 / 256)
        loss = F.mse_loss(res["rgb"], target)
        if w_dist > 0:
            loss = loss + w_dist * distortion_loss(res).mean()              # train.py:194-195
    (loss * 256.0).backward()
    os.environ["NGP_FUSED_RENDER"] = "1"
    m.use_fused_mlp = True
    return res, loss.item(), [p.grad.clone().float() for p in [m.pos_encoder.hash_table, *m._mlp_weights()]]


def test_garden_config_fused_matches_operator_path_with_distortion(hip_lib):
    m, o, d, target = _garden()
    r_f, l_f, g_f = _step(m, o, d, target, True, 1e-3)
    r_o, l_o, g_o = _step(m, o, d, target, False, 1e-3)
    assert torch.equal(r_f["rays_a"][:, [0, 2]], r_o["rays_a"][:, [0, 2]]) and int(r_f["rm_samples"]) == int(r_o["rm_samples"]) > 10000
    torch.testing.assert_close(r_f["rgb"], r_o["rgb"], rtol=0, atol=5e-3)
    assert abs(l_f - l_o) < 2e-3
    for a, b in zip(g_f, g_o):
        assert ((a - b).norm() / b.norm().clamp_min(1e-30)).item() < 5e-2
    # black background for real scenes (rendering.py:219-226): rays that hit nothing stay black
    empty = r_f["opacity"] == 0
    assert empty.any() and float(r_f["rgb"][empty].abs().max()) == 0.0


def test_garden_trainer_with_distortion_loss_gradients(hip_lib):
    """FusedTrainer(distortion_loss_w > 0) -- composite fwd, distortion fwd/bwd, MSE gradient, composite bwd -- against the
    reference-shaped autograd path (render() + F.mse_loss + w * distortion_loss().mean(), train.py:193-195)."""
    from ngp_hip.trainer import FusedTrainer
    w = 1e-2
    m, o, d, target = _garden()
    _, _, g_ref = _step(m, o, d, target, False, w)                            # gradients of 256 * loss
    _, _, g_ref0 = _step(m, o, d, target, False, 0.0)
    tr = FusedTrainer(m, exp_step_factor=1 / 256, distortion_loss_w=w, init_scale=256.0)
    torch.manual_seed(7)
    out = tr.compute_gradients(o, d, target, noise=torch.rand(o.shape[0], device="cuda"))     # the jitter _step()'s render() drew
    assert int(out["found_inf"]) == 0
    got = [out["table_grad"] * 256.0] + [g.reshape(-1) * 256.0 for g in torch.split(out["mlp_grad"], [2048, 1024, 2048, 4096, 192])]
    for k, (a, b, b0) in enumerate(zip(got, g_ref, g_ref0)):
        b, b0 = b.reshape(-1), b0.reshape(-1)
        err = ((a - b).norm() / b.norm()).item()
        assert err < 5e-2
        if k < 3:       # the sample weights depend on the densities only: table and density-MLP gradients carry the distortion term
            assert ((b - b0).norm() / b.norm()).item() > 5 * err
        else:           # ... and the colour MLP's do not, exactly
            assert torch.equal(b, b0)
    # and a few full steps run (loss decreases, nothing skipped for lack of finiteness)
    l0 = None
    for i in range(8):
        tr.step(o, d, target)
        l0 = tr.last_loss() if l0 is None else l0
    assert tr.last_loss() < l0 and tr.counters()["skipped"] == 0


def test_garden_eval_path_matches_train_path_without_jitter(hip_lib):
    """Appendix B.9: progressive raymarching_test + composite_test == one-shot train-style composite."""
    from modules.rendering import render
    m, o, d, _ = _garden(2048)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        test = render(m, o, d, test_time=True, exp_step_factor=1 / 256)
    # train-style render with the jitter forced to zero
    import modules.ray_march as rm
    orig = torch.rand_like
    try:
        torch.rand_like = lambda x, *a, **k: torch.zeros_like(x)
        import os
        os.environ["NGP_FUSED_RENDER"] = "0"
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            train = render(m, o, d, exp_step_factor=1 / 256)
    finally:
        torch.rand_like = orig
        os.environ["NGP_FUSED_RENDER"] = "1"
    torch.testing.assert_close(test["rgb"], train["rgb"], rtol=0, atol=3e-3)
    torch.testing.assert_close(test["opacity"], train["opacity"], rtol=0, atol=3e-3)


def test_half_encoder_model_trains(hip_lib, lego_bitfield):
    """C5: NGP(half_opt=True) -- fp16 table copy, f16 gather, packed-f16 atomic backward -- through the reference's
    loop shape (GradScaler 2^16, train.py:137-141)."""
    from modules.networks import NGP
    from modules.rendering import render
    from ngp_hip import synthetic
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024, half_opt=True).cuda()
    assert m.pos_encoder.hash_table.shape == (5710032, 2) and "pos_encoder.hash_grad" in m.state_dict()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    o, d = synthetic.lego_rays(2048, seed=1)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(2048, 1, device="cuda").expand(-1, 3) * 0.5
    opt = torch.optim.Adam(m.parameters(), 1e-2, eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0**16)
    losses = []
    for _ in range(120):
        with torch.autocast("cuda", dtype=torch.float16):
            res = render(m, o, d, exp_step_factor=0.0)
            loss = F.mse_loss(res["rgb"], target)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < 0.85 * losses[0], (losses[0], losses[-1])
    assert m.pos_encoder.hash_table.grad is not None and m.pos_encoder.hash_table.grad.dtype == torch.float32


def test_density_grid_update_and_mark_invisible(hip_lib):
    """Occupancy maintenance (networks.py:168-290) on the HIP kernels: warm-up and sampled updates, bitfield == packbits."""
    from modules.networks import NGP
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024).cuda()
    K = torch.tensor([[277.8, 0, 100.0], [0, 277.8, 100.0], [0, 0, 1]], device="cuda")
    poses = torch.eye(4, device="cuda")[:3].repeat(4, 1, 1)
    poses[:, :, 3] = torch.tensor([[0, 0, -1.3], [0.2, 0, -1.3], [0, 0.2, -1.3], [-0.2, 0, -1.3]], device="cuda")
    m.mark_invisible_cells(K, poses, (200, 200))
    assert (m.density_grid < 0).any() and (m.density_grid == 0).any()
    with torch.autocast("cuda", dtype=torch.float16):
        m.update_density_grid(5.9, warmup=True)
        g1 = m.density_grid.clone()
        m.update_density_grid(5.9, warmup=False)
    assert (m.density_grid[g1 < 0] < 0).all()                         # invisible cells stay marked
    thr = min(m.density_grid[m.density_grid > 0].mean().item(), 5.9)
    want = (m.density_grid.reshape(-1, 8) > thr).to(torch.uint8)
    want = (want * (2 ** torch.arange(8, device="cuda", dtype=torch.uint8))).sum(1).to(torch.uint8)
    assert torch.equal(want, m.density_bitfield)
    assert 0.05 < (m.density_grid > thr).float().mean().item() < 0.95


def test_fused_occupancy_update_matches_torch_formulation(hip_lib, monkeypatch):
    """ngp_hip/occupancy.py vs the torch-op formulation of the same algorithm (networks.py:255-290): with the jitter
    pinned to the cell centre the warm-up update is deterministic in both."""
    import copy
    from modules.networks import NGP
    torch.manual_seed(0)
    m_a = NGP(scale=0.5, max_res=1024).cuda()
    with torch.no_grad():
        m_a.density_grid[0, ::7] = -1.0                                    # some cells marked invisible
    m_b = copy.deepcopy(m_a)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.full(a if not isinstance(a[0], (tuple, list)) else tuple(a[0]), 0.5,
                                                                  device=k.get("device")))
    monkeypatch.setattr(torch, "rand_like", lambda x, *a, **k: torch.full_like(x, 0.5))
    with torch.autocast("cuda", dtype=torch.float16):
        monkeypatch.setenv("NGP_FUSED_OCCUPANCY", "1")
        m_a.update_density_grid(5.9, warmup=True)
        monkeypatch.setenv("NGP_FUSED_OCCUPANCY", "0")
        m_b.update_density_grid(5.9, warmup=True)
    assert (m_a.density_grid[0, ::7] == -1).all()
    torch.testing.assert_close(m_a.density_grid, m_b.density_grid, rtol=2e-3, atol=1e-4)
    agree = (m_a.density_bitfield == m_b.density_bitfield).float().mean().item()
    assert agree > 0.995, agree
    # sampled update: invariants of the algorithm
    monkeypatch.undo()
    before = m_a.density_grid.clone()
    with torch.autocast("cuda", dtype=torch.float16):
        m_a.update_density_grid(5.9, warmup=False)
    g = m_a.density_grid
    assert (g[before < 0] == before[before < 0]).all()
    assert (g[before >= 0] >= before[before >= 0] * 0.95 - 1e-6).all()      # decay/max merge
    assert ((g - before * 0.95).abs() > 1e-6).float().mean().item() > 0.2   # a good part of the cells was re-sampled


def test_occupancy_compaction_is_deterministic_and_ordered(hip_lib):
    """ngp_occ_compact: list == the occupied cells in cell order (what torch.nonzero gives in networks.py:198-199), count exact,
    identical on every call (replicas on different ranks pick the same cells from it)."""
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    torch.manual_seed(4)
    for n_cells, frac in ((128**3, 0.04), (128**3, 0.9), (100003, 0.5), (64, 1.0)):
        grid = torch.rand(n_cells, device="cuda")
        thr = 1.0 - frac
        lst = torch.full((n_cells,), -1, device="cuda", dtype=torch.int32)
        cnt = torch.full((1,), -7, device="cuda", dtype=torch.int32)
        scratch = torch.empty(1024, device="cuda", dtype=torch.int32)
        L.check(lib.ngp_occ_compact(_ptr(grid), thr, n_cells, _ptr(lst), _ptr(cnt), _ptr(scratch), _stream()), "ngp_occ_compact")
        want = torch.nonzero(grid > thr)[:, 0].to(torch.int32)
        assert int(cnt) == want.numel()
        assert torch.equal(lst[:want.numel()], want) and (lst[want.numel():] == -1).all()


def test_sorted_uniforms_are_order_statistics(hip_lib):
    """ngp_sorted_uniforms: strictly inside (0, 1), ascending, and distributed like sorted iid uniforms (k-th value ~ k / (m + 1),
    the spacing fluctuations of a uniform sample) -- what the occupancy update feeds ngp_occ_sample (networks.py:193-203 draws
    iid random cells; sorting them only changes the order of the encoder queries)."""
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    torch.manual_seed(3)
    for m in (524288, 1000, 1023, 1024):
        rows = (m + 1 + 1023) // 1024
        u = torch.rand(2 * rows * 1024, device="cuda")
        work = torch.empty(2 * (rows * 1024 + rows), device="cuda")
        out = torch.empty(2, m, device="cuda")
        L.check(lib.ngp_sorted_uniforms(_ptr(u), m, 2, _ptr(work), _ptr(out), _stream()), "ngp_sorted_uniforms")
        assert float(out.min()) > 0.0 and float(out.max()) < 1.0
        assert bool((out[:, 1:] >= out[:, :-1]).all())
        k = torch.arange(1, m + 1, device="cuda", dtype=torch.float64) / (m + 1)
        dev = (out.double() - k).abs().max().item()
        assert dev < 4.0 / m**0.5, (m, dev)                      # Kolmogorov-Smirnov scale: sup |F_m - F| ~ 1 / sqrt(m)
        assert not torch.equal(out[0], out[1])
        # the closed form, in double
        e = -torch.log1p(-u.double()).view(2, -1)
        cs = torch.cumsum(e, 1)
        want = cs[:, :m] / cs[:, m:m + 1]
        torch.testing.assert_close(out.double(), want, rtol=2e-4, atol=2e-6)
