"""FusedTrainer (device-resident loss / GradScaler / Adam / cosine LR, persistent gradients) against the reference's
own loop shape -- render() through modules/ + torch.optim.Adam(eps=1e-15) + torch GradScaler + CosineAnnealingLR
(train.py:137-201) -- on identical rays, targets and jitter noise."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _make(lego_bitfield, n=4096):
    from modules.networks import NGP
    from ngp_hip import synthetic
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    o, d = synthetic.lego_rays(n, seed=9)
    return m, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.rand(n, 3, device="cuda")


def test_trainer_matches_torch_loop(hip_lib, lego_bitfield):
    from modules.rendering import render
    from ngp_hip.trainer import FusedTrainer
    m_a, o, d, target = _make(lego_bitfield)
    m_b = copy.deepcopy(m_a)
    steps, T = 6, 50
    # (a) torch loop, as train.py builds it
    opt = torch.optim.Adam(m_a.parameters(), 1e-2, eps=1e-15)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T, 1e-2 / 30)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0**19)
    losses_a = []
    for i in range(steps):
        torch.manual_seed(100 + i)
        with torch.autocast("cuda", dtype=torch.float16):
            res = render(m_a, o, d, exp_step_factor=0.0)
            loss = F.mse_loss(res["rgb"], target)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        sched.step()
        losses_a.append(loss.item())
    # (b) fused trainer
    tr = FusedTrainer(m_b, lr=1e-2, max_steps=T, init_scale=2.0**19)
    losses_b = []
    for i in range(steps):
        torch.manual_seed(100 + i)
        st = tr.step(o, d, target, noise=torch.rand(o.shape[0], device="cuda"))       # the jitter render() drew for this seed
        losses_b.append(tr.last_loss())
    assert tr.counters()["iter"] == steps
    # both loops skip the same (early, overflowing) steps or none; compare the trajectories
    np.testing.assert_allclose(losses_b, losses_a, rtol=2e-2, atol=2e-3)
    assert tr.counters()["opt_steps"] == steps - tr.counters()["skipped"]
    if tr.counters()["skipped"] == 0 and scaler.get_scale() == 2.0**19:
        ta, tb = m_a.pos_encoder.hash_table.detach(), m_b.pos_encoder.hash_table.detach()
        moved = (ta - copy.deepcopy(_make(lego_bitfield)[0]).pos_encoder.hash_table.detach()).abs() > 0
        assert moved.any()
        rel = ((ta - tb)[moved].norm() / (ta[moved].norm())).item()
        assert rel < 2e-2, rel
        for wa, wb in zip(m_a._mlp_weights(), m_b._mlp_weights()):
            assert ((wa - wb).norm() / wa.norm()).item() < 5e-2
    assert losses_b[-1] < losses_b[0]


def test_trainer_inf_skips_and_backs_off(hip_lib, lego_bitfield):
    from ngp_hip.trainer import FusedTrainer
    m, o, d, target = _make(lego_bitfield, n=1024)
    tr = FusedTrainer(m, init_scale=2.0**40)                  # guaranteed fp16 overflow in the first steps
    before = m.pos_encoder.hash_table.detach().clone()
    tr.step(o, d, target)
    c = tr.counters()
    assert c["skipped"] == 1 and c["opt_steps"] == 0 and tr.loss_scale() == 2.0**39
    assert torch.equal(before, m.pos_encoder.hash_table.detach())          # a skipped step leaves the parameters alone
    assert float(tr.table_grad.abs().max()) == 0.0                        # ... and still clears the gradients
    for _ in range(40):
        tr.step(o, d, target)
    c = tr.counters()
    assert c["opt_steps"] > 0 and np.isfinite(tr.last_loss())


def test_trainer_graph_replay_equals_eager(hip_lib, lego_bitfield):
    """hipGraph mode (shading/backward/optimizer chain as one graph launch, marches eager / on the side stream) against the eager
    trainer: same counters, same sample counts (up to the jitter noise), same loss trajectory -- across prefetched and
    non-prefetched steps and across a bitfield change (the 8^3-block occupancy shortcut must follow it)."""
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    n = 2048
    m_a, o, d, target = _make(lego_bitfield, n=n)
    m_b = copy.deepcopy(m_a)
    tr_a = FusedTrainer(m_a, init_scale=2.0**10)
    tr_b = FusedTrainer(m_b, init_scale=2.0**10)
    pool = [(o, d)] + [tuple(torch.from_numpy(x).cuda() for x in synthetic.lego_rays(n, seed=40 + k)) for k in range(2)]
    tr_b.capture(n)
    ones = torch.full_like(m_a.density_bitfield, 255)
    counts_a, counts_b = [], []
    for i in range(9):
        if i == 5:                                   # occupancy change between steps: no prefetch across it
            for m in (m_a, m_b):
                m.density_bitfield.copy_(ones)
        ro, rd = pool[i % 3]
        nxt = pool[(i + 1) % 3]
        pre = None if (i + 1 == 5 or i % 4 == 3) else nxt       # some steps without lookahead, never across the change
        sa = tr_a.step(ro, rd, target, prefetch=pre)
        sb = tr_b.step(ro, rd, target, prefetch=pre)
        counts_a.append(int(sa["rm_samples"][0])); counts_b.append(int(sb["rm_samples"][0]))
    torch.cuda.synchronize()
    assert tr_a.counters() == tr_b.counters()
    assert len(tr_b._graph) == 2                     # one graph per march-set parity
    for ca, cb in zip(counts_a, counts_b):           # jitter differs (graph-safe philox offsets): equal up to a few samples per ray
        assert abs(ca - cb) <= 0.03 * max(ca, cb) + 64, (counts_a, counts_b)
    assert min(counts_a[5:]) > 5 * max(counts_a[:5])             # the all-ones grid really took effect in both modes
    assert abs(tr_a.last_loss() - tr_b.last_loss()) < 2e-2


def test_trainer_allreduce_path_on_rccl_world1(hip_lib, lego_bitfield):
    """The N>1 exchange step (ONE RCCL all-reduce of the flat bucket [table grad | MLP grad | inf flag]) exercised on the real `nccl`
    backend with a 1-rank group: averaging over one rank must leave every buffer bit-identical.  (Multi-rank numerics are
    covered on CPU by tests/test_dist_gloo.py; the 8-GPU run is the driver's.)"""
    import os
    import socket
    import torch.distributed as dist
    from ngp_hip.trainer import FusedTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m, o, d, target = _make(lego_bitfield, n=1024)
        tr = FusedTrainer(m, init_scale=2.0**10, world_size=1)
        tr.table_grad.normal_(); tr.mlp_grad.normal_()
        tg, mg = tr.table_grad.clone(), tr.mlp_grad.clone()
        tr.world = 2                      # force the exchange code path; AVG over the single rank is the identity
        tr._all_reduce()
        torch.cuda.synchronize()
        assert torch.equal(tg, tr.table_grad) and torch.equal(mg, tr.mlp_grad)
        # the inf flag rides in the same bucket: raised on this rank -> still raised after the collective, slot cleared
        tr.state_i[3] = 1
        tr._all_reduce()
        assert int(tr.state_i[3]) == 1 and float(tr._flag_f) == 0.0
        tr.state_i[3] = 0
        tr._all_reduce()
        assert int(tr.state_i[3]) == 0
        # optional bf16 transport: gradients come back rounded to bf16 (rel. error <= 2^-8), nothing else changes
        tr._comm = torch.empty_like(tr.grad_flat, dtype=torch.bfloat16)
        tr._all_reduce()
        torch.cuda.synchronize()
        assert torch.equal(tr.table_grad, tg.bfloat16().float()) and torch.equal(tr.mlp_grad, mg.bfloat16().float())
        tr._comm = None
        tr.table_grad.zero_(); tr.mlp_grad.zero_()
        for _ in range(3):
            tr.step(o, d, target)         # full steps with the collective inside
        torch.cuda.synchronize()
        assert tr.counters()["iter"] == 3 and np.isfinite(tr.last_loss())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["f32", "bf16", "f32-overlap"])
def test_sharded_optimizer_on_rccl_world1(hip_lib, lego_bitfield, kind, monkeypatch):
    """reduce-scatter -> Adam on the own shard -> all-gather (the N > 1 default) on the real `nccl` backend with a 1-rank group:
    the shard is the whole table, so three steps must land where the plain single-GPU trainer lands (same kernels; only the
    float-atomic flush order of the replicated coarse levels differs run to run).  "f32-overlap": the same with the exchange
    split by level group and issued async under the scatter-add (NGP_EXPERIMENT comm_overlap=1), i.e. RCCL's async reduce-scatter /
    all-gather on the group staging buffers."""
    import os
    overlap = kind == "f32-overlap"
    if overlap:
        monkeypatch.setenv("NGP_EXPERIMENT", "comm_overlap=1;comm_groups=12,8,0")
    import socket
    import torch.distributed as dist
    from modules.networks import NGP
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        def make():
            torch.manual_seed(0)
            m = NGP(scale=0.5, max_res=1024, table_dtype=torch.bfloat16 if kind == "bf16" else None).cuda()
            m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
            with torch.no_grad():
                m.pos_encoder.hash_table.mul_(0.2)
            return m
        o, d = [torch.from_numpy(a).cuda() for a in synthetic.lego_rays(2048, seed=9)]
        target = torch.rand(2048, 3, device="cuda")
        tr_a = FusedTrainer(make(), world_size=1, shard_optimizer=True, init_scale=2.0**15)
        monkeypatch.delenv("NGP_EXPERIMENT", raising=False)
        tr_b = FusedTrainer(make(), world_size=1, init_scale=2.0**15)
        assert tr_a.shard and not tr_b.shard and tr_a.shard_len == tr_a.nt_pad
        assert (tr_a._groups is not None) == overlap and tr_b._groups is None
        for i in range(3):
            noise = torch.rand(2048, device="cuda")
            tr_a.step(o, d, target, noise=noise); tr_b.step(o, d, target, noise=noise)
        tr_a.sync_master()
        torch.cuda.synchronize()
        assert tr_a.counters() == tr_b.counters() and tr_a.counters()["opt_steps"] == 3
        assert ((tr_a.table - tr_b.table).norm() / tr_b.table.norm()).item() < 1e-6
        assert ((tr_a.mlp_flat - tr_b.mlp_flat).norm() / tr_b.mlp_flat.norm()).item() < 1e-6
        if kind == "bf16":
            assert torch.equal(tr_a.table.bfloat16().view(torch.int16), tr_a.copy16_store[:tr_a.nt].view(torch.int16))
        assert float(tr_a.table_grad_store.abs().max()) == 0.0 and float(tr_a.shard_grad.abs().max()) == 0.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True])
def test_adam_all_equals_separate_launches(hip_lib, bf16):
    """ngp_adam_all (table pass + MLP Adam + fragment repack in one launch) == ngp_adam_step[_bf16] + ngp_adam_mlp_pack,
    bit for bit, including the skipped-step branch."""
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    n = 1 << 20
    for skip in (0, 1):
        torch.manual_seed(3 + skip)
        mk = lambda k: torch.randn(k, device="cuda")
        tp, tg, tm, tv = mk(n), mk(n) * 512, mk(n).abs() * 0.1, mk(n).abs() * 0.01
        tg[: n // 4] = 0; tm[: n // 4] = 0; tv[: n // 4] = 0                  # never-touched entries
        wp, wg, wm, wv = mk(9408) * 0.2, mk(9408) * 512, mk(9408) * 0.1, mk(9408).abs() * 0.01
        sf = torch.zeros(8, device="cuda"); si = torch.zeros(8, device="cuda", dtype=torch.int32)
        sf[0] = 512.0; si[3] = skip
        L.check(lib.ngp_train_prologue(_ptr(sf), _ptr(si), 1e-2, 1e-2 / 30, 100, 0.9, 0.999, 2.0, 0.5, 2000, _stream()), "prologue")
        A = [t.clone() for t in (tp, tg, tm, tv, wp, wg, wm, wv)]
        B = [t.clone() for t in (tp, tg, tm, tv, wp, wg, wm, wv)]
        sh_a = A[0].bfloat16() if bf16 else None
        sh_b = B[0].bfloat16() if bf16 else None
        nh = lib.ngp_mlp_wpack_halfs()
        wk_a, wk_b = torch.zeros(nh, device="cuda", dtype=torch.float16), torch.zeros(nh, device="cuda", dtype=torch.float16)
        L.check(lib.ngp_adam_all(_ptr(A[0]), _ptr(A[1]), _ptr(A[2]), _ptr(A[3]), n, _ptr(sh_a), _ptr(A[4]), _ptr(A[5]), _ptr(A[6]),
                                 _ptr(A[7]), _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, 1, _ptr(wk_a), _stream()), "adam_all")
        if bf16:
            L.check(lib.ngp_adam_step_bf16(_ptr(B[0]), _ptr(B[1]), _ptr(B[2]), _ptr(B[3]), n, _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15,
                                           _ptr(sh_b), _stream()), "adam_bf16")
        else:
            L.check(lib.ngp_adam_step(_ptr(B[0]), _ptr(B[1]), _ptr(B[2]), _ptr(B[3]), n, _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15,
                                      _stream()), "adam")
        L.check(lib.ngp_adam_mlp_pack(_ptr(B[4]), _ptr(B[5]), _ptr(B[6]), _ptr(B[7]), _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, 1,
                                      _ptr(wk_b), _stream()), "adam_mlp_pack")
        torch.cuda.synchronize()
        for a, b in zip(A, B):
            assert torch.equal(a, b)
        assert torch.equal(wk_a.view(torch.int16), wk_b.view(torch.int16)) and wk_a.any()
        if bf16:
            assert torch.equal(sh_a.view(torch.int16), sh_b.view(torch.int16))
        assert (skip == 1) == torch.equal(A[0], tp)                            # a skipped step leaves the parameters alone


def test_live_backward_equals_full_backward(hip_lib, lego_bitfield):
    """Backward over the compacted live-sample list (ngp_live_compact + *_live kernels) == backward over every marched sample:
    the skipped samples sit behind their ray's early-termination point and carry exact-zero gradients."""
    from ngp_hip.trainer import FusedTrainer
    m, o, d, target = _make(lego_bitfield, n=4096)
    tr = FusedTrainer(m, init_scale=2.0**10)
    # raise the density until a good part of the rays terminates early (exp(h0) has to reach ~1e4 at this step size)
    for _ in range(10):
        st = tr.compute_gradients(o, d, target)
        if int(st["vr_per_ray"].sum()) < 0.6 * int(st["rm_samples"][0]):
            break
        with torch.no_grad():
            m.xyz_encoder.output_layer.weight.mul_(3.0)
        tr.repack()
    from ngp_hip.fused import TrainArena
    outs = []
    for live in (True, False):
        tr.live_backward = live
        torch.manual_seed(77)
        outs.append(tr.compute_gradients(o, d, target))
        if live:
            n_live = int(tr._live_total)
            got_list = TrainArena.get(o.device, o.shape[0], 1024).live_idx[:n_live].cpu().numpy().copy()
    a, b = outs
    n_all = int(a["rm_samples"][0])
    assert 0 < n_live < 0.8 * n_all and n_live == int(a["vr_per_ray"].sum())
    # (same samples per ray; where a ray's range starts is decided by the order the march kernel's blocks finish in)
    assert torch.equal(a["rays_a"][:, [0, 2]], b["rays_a"][:, [0, 2]]) and torch.equal(a["rgb"], b["rgb"])
    for k in ("table_grad", "mlp_grad"):
        ga, gb = a[k], b[k]
        assert torch.equal(ga != 0, gb != 0) or ((ga != 0) == (gb != 0)).float().mean().item() > 0.9999
        assert ((ga - gb).norm() / gb.norm()).item() < 1e-5      # float-atomic / MFMA accumulation order only
    # the live list is the per-ray prefixes: every ray's run contiguous and ascending, the rays in the order their 16-ray blocks
    # of the fused composite kernel finished (ngp_composite_train_fused_live), i.e. a permutation of the ray-order list
    ra, vr = a["rays_a"].cpu().numpy(), a["vr_per_ray"].cpu().numpy()
    want = np.sort(np.concatenate([np.arange(s, s + vr[r]) for r, s, c in ra])) if n_live else np.zeros(0)
    assert np.array_equal(np.sort(got_list), want)
    run_start = {int(s): int(vr[r]) for r, s, c in ra if vr[r] > 0}
    pos = 0
    while pos < n_live:
        length = run_start[int(got_list[pos])]
        assert np.array_equal(got_list[pos:pos + length], np.arange(got_list[pos], got_list[pos] + length))
        pos += length
    # the operator (two-pass scan + fill, ray order) over the same per-ray counts
    from ngp_hip import lib as _lib
    from ngp_hip.ops import _ptr, _stream
    L = _lib.load()
    n_rays = ra.shape[0]
    off = torch.empty(n_rays, dtype=torch.int32, device="cuda"); lst = torch.empty(n_all, dtype=torch.int32, device="cuda")
    tot = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert L.ngp_live_compact(_ptr(a["rays_a"]), _ptr(a["vr_per_ray"]), n_rays, _ptr(off), _ptr(lst), _ptr(tot), _stream()) == 0
    want_ray_order = np.concatenate([np.arange(s, s + vr[r]) for r, s, c in ra])
    assert int(tot) == n_live and np.array_equal(lst[:n_live].cpu().numpy(), want_ray_order)


@pytest.mark.parametrize("kind", ["f32", "bf16", "half"])
def test_checkpoint_roundtrip_through_trainer(hip_lib, lego_bitfield, kind):
    """state_dict out of a trained FusedTrainer model -> fresh model + trainer (load_state_dict + repack) renders the same image;
    the 16-bit table copies and the packed MLP fragments follow the loaded parameters."""
    from modules.networks import NGP
    from modules.rendering import render
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    kw = dict(table_dtype=torch.bfloat16) if kind == "bf16" else (dict(half_opt=True) if kind == "half" else {})
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024, **kw).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.uniform_(0.0, 0.2)
    o, d = synthetic.lego_rays(2048, seed=9)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(2048, 3, device="cuda")
    tr = FusedTrainer(m, init_scale=2.0**7)
    for _ in range(5):
        tr.step(o, d, target)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = NGP(scale=0.5, max_res=1024, **kw).cuda()
    tr2 = FusedTrainer(m2, init_scale=2.0**7)
    m2.load_state_dict(sd)
    tr2.repack()
    for mm in (m, m2):
        mm.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = render(m, o, d, test_time=True, exp_step_factor=0.0)["rgb"]
        b = render(m2, o, d, test_time=True, exp_step_factor=0.0)["rgb"]
    assert torch.equal(a, b)
    if kind == "bf16":
        assert torch.equal(tr2.table_bf16.view(torch.int16), m2.pos_encoder.hash_table.detach().bfloat16().view(torch.int16))
    if kind == "half":
        assert torch.equal(tr2.table_f16.view(torch.int16), m2.pos_encoder.hash_table.detach().reshape(-1).half().view(torch.int16))
    # and training continues from the loaded state (the flat MLP buffer still aliases the module parameters)
    l0 = None
    for _ in range(4):
        tr2.step(o, d, target)
        l0 = tr2.last_loss() if l0 is None else l0
    assert np.isfinite(tr2.last_loss()) and tr2.last_loss() <= l0 * 1.05


def test_coarse_table_follows_grid_update(hip_lib, lego_bitfield):
    """the 8^3-block coarse occupancy table must be rebuilt after every writer of the bitfield, including the
    raw-pointer kernels (ngp_occ_pack / ngp_packbits) -- a stale table vetoes cells that became occupied.  Start from an EMPTY
    bitfield (coarse table all zero), let the trainer's own update fill it, and check the next step's march against the
    operator path on the same rays and noise."""
    from modules.networks import NGP
    from modules.ray_march import raymarching_train
    from modules.intersection import ray_aabb_intersection
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024).cuda()
    o, d = synthetic.lego_rays(2048, seed=4)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(2048, 3, device="cuda")
    tr = FusedTrainer(m)
    st = tr.step(o, d, target)                                   # empty bitfield: coarse table = 0, nothing marched
    assert int(st["rm_samples"][0]) == 0
    tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=True)    # writes the bitfield through ngp_occ_pack (raw pointer)
    assert int(m.density_bitfield.count_nonzero()) > 0
    torch.manual_seed(77)
    st = tr.step(o, d, target, noise=torch.rand(2048, device="cuda"))   # (the trainer draws its own jitter in the march kernel otherwise)
    torch.manual_seed(77)                                        # same jitter noise: first draw after the seed
    hits = ray_aabb_intersection(o, d, 0.5)
    rays_a, _x, _d, _dl, _ts, total = raymarching_train(o, d, hits, m.density_bitfield, 1, 0.5, 0.0, 128, 1024)
    assert int(st["rm_samples"][0]) == int(total) > 0
    assert torch.equal(st["rays_a"][:, [0, 2]], rays_a[:, [0, 2]])          # (the trainer's march packs the rays in block order)
    # ... and through modules.utils.packbits (ngp_packbits) as well
    from modules.utils import packbits
    v0 = m.density_bitfield._version
    packbits(m.density_grid.reshape(-1).contiguous(), 1e9, m.density_bitfield)          # threshold above everything: empties it
    assert m.density_bitfield._version > v0
    st = tr.step(o, d, target)
    assert int(st["rm_samples"][0]) == 0


def test_stale_prefetch_is_not_consumed(hip_lib, lego_bitfield):
    """a prefetched march is used only for the very tensors it was issued for, unmodified; anything else
    re-marches (after waiting for the side stream) and gives the un-prefetched result."""
    from ngp_hip.trainer import FusedTrainer
    from ngp_hip import synthetic
    m, o, d, target = _make(lego_bitfield, n=2048)
    o2, d2 = [torch.from_numpy(a).cuda() for a in synthetic.lego_rays(2048, seed=10)]
    o3, d3 = [torch.from_numpy(a).cuda() for a in synthetic.lego_rays(2048, seed=11)]
    tr = FusedTrainer(m)
    tr._grads_only = True                                        # keep the parameters fixed: only the march matters here
    torch.manual_seed(5)
    tr._launch(o, d, target, (o2, d2), (o, d), (o2, d2))         # prefetches (o2, d2) ...
    torch.manual_seed(6)
    got = tr._launch(o3, d3, target, None, (o3, d3), None)       # ... but the next step brings other rays
    ra = got["rays_a"].clone(); rm = int(got["rm_samples"][0])
    tr2 = FusedTrainer(_make(lego_bitfield, n=2048)[0]); tr2._grads_only = True
    torch.manual_seed(6)
    ref = tr2._launch(o3, d3, target, None, (o3, d3), None)
    assert rm == int(ref["rm_samples"][0]) and torch.equal(ra[:, [0, 2]], ref["rays_a"][:, [0, 2]])
    # modified in place after the prefetch -> version moved -> not consumed
    torch.manual_seed(5)
    tr._launch(o, d, target, (o2, d2), (o, d), (o2, d2))
    o2.add_(0.0)
    M = tr._march_sets(2048)[tr._cur]
    assert M.ready is not None and not M.marched_for((o2, d2))
    tr._grads_only = False


def test_fused_render_backward_after_second_forward_raises(hip_lib, lego_bitfield):
    """the fused render keeps activations in a shared arena; differentiating a forward whose
    activations were overwritten must fail loudly, never silently use the other batch's activations."""
    from modules.rendering import render
    m, o, d, target = _make(lego_bitfield, n=1024)
    with torch.autocast("cuda", dtype=torch.float16):
        r1 = render(m, o, d, exp_step_factor=0.0)
        r2 = render(m, o, d, exp_step_factor=0.0)
        l1, l2 = F.mse_loss(r1["rgb"], target), F.mse_loss(r2["rgb"], target)
    l2.backward()                                                # latest forward: fine
    with pytest.raises(RuntimeError, match="overwritten by a later render"):
        l1.backward()


def test_trainer_state_dict_resumes_exactly(hip_lib, lego_bitfield):
    from ngp_hip.trainer import FusedTrainer
    m_a, o, d, target = _make(lego_bitfield, n=1024)
    tr_a = FusedTrainer(m_a, max_steps=100)
    for i in range(4):
        torch.manual_seed(i); tr_a.step(o, d, target)
    sd_model, sd_opt = copy.deepcopy(m_a.state_dict()), tr_a.state_dict()
    for i in range(4, 7):
        torch.manual_seed(i); tr_a.step(o, d, target)
    m_b = _make(lego_bitfield, n=1024)[0]
    tr_b = FusedTrainer(m_b, max_steps=100)
    m_b.load_state_dict(sd_model)
    tr_b.load_state_dict(sd_opt)
    for i in range(4, 7):
        torch.manual_seed(i); tr_b.step(o, d, target)
    assert tr_a.counters() == tr_b.counters()
    # same kernels, same inputs; only the float-atomic order of the scatter-add differs between the two runs
    ta, tb = m_a.pos_encoder.hash_table.detach(), m_b.pos_encoder.hash_table.detach()
    assert ((ta - tb).norm() / ta.norm()).item() < 1e-4


def test_step_after_a_failed_launch_does_not_inherit_live_counts(hip_lib, lego_bitfield):
    """a _launch() that raises after the march-set parity flip leaves the NEXT step of that parity with a non-zero
    live counter (the composite kernel that clears it never ran) -- the trainer now clears the counters when the previous launch
    did not complete.  Provoked by a failing entry point in the middle of the step."""
    from ngp_hip.trainer import FusedTrainer
    m, o, d, target = _make(lego_bitfield, n=2048)
    tr = FusedTrainer(m)
    tr.step(o, d, target)
    tr.step(o, d, target)
    ref_live = int(tr._live_total[0])

    class Boom(Exception):
        pass

    class Flaky:
        def __init__(self, L):
            self._L, self.fail = L, False

        def __getattr__(self, name):
            fn = getattr(self._L, name)
            if name == "ngp_mlp_fwd_ex" and self.fail:
                def boom(*a):
                    raise Boom()
                return boom
            return fn
    tr.L = Flaky(tr.L)
    tr._live_pair.fill_(12345)                                   # what stale counters would look like
    tr.L.fail = True
    with pytest.raises(Boom):
        tr.step(o, d, target)
    tr.L.fail = False
    tr.step(o, d, target)
    assert abs(int(tr._live_total[0]) - ref_live) < 0.2 * ref_live + 64           # a fresh count, not 12345 + count
    assert tr.counters()["skipped"] == 0


def test_train_results_mapping_protocol(hip_lib, lego_bitfield):
    """pop / setdefault / copy / dict() of the fused render's result see the lazily materialised per-sample keys."""
    import copy as _copy
    from modules.rendering import render
    m, o, d, _ = _make(lego_bitfield, n=1024)
    with torch.autocast("cuda", dtype=torch.float16):
        res = render(m, o, d, exp_step_factor=0.0)
    S = int(res["rm_samples"])
    plain = res.copy()
    assert type(plain) is dict and plain["ws"].shape[0] == S and set(plain) == set(res.keys())
    assert dict(res)["ts"].shape[0] == S
    assert res.setdefault("deltas", None).shape[0] == S and res.setdefault("extra", 3) == 3
    ws = res.pop("ws")
    assert ws.shape[0] == S and "ws" not in res and res.pop("ws", "gone") == "gone"
    assert type(_copy.deepcopy({k: v.detach() if torch.is_tensor(v) else v for k, v in res.copy().items()})) is dict


def test_mlp_slab_sum_in_scatter_launch_equals_other_paths(hip_lib, lego_bitfield, monkeypatch):
    """Round 4: the MLP backward's per-block weight-gradient slabs are summed by the head of the scatter-add launch
    (ngp_hash_bwd_sliced_main_slabs), by the prologue launch (NGP_EXPERIMENT mlp_dw_reduce=prologue), or not used at all (mlp_dw=atomic:
    round 3's float atomics): three trainers from the same model and jitter take the same first steps."""
    from ngp_hip.trainer import FusedTrainer
    outs = []
    for env in ("mlp_dw_reduce=scatter", "mlp_dw_reduce=prologue", "mlp_dw=atomic"):
        monkeypatch.setenv("NGP_EXPERIMENT", env)
        m, o, d, target = _make(lego_bitfield, n=2048)
        tr = FusedTrainer(m, init_scale=2.0**12)
        g = torch.Generator(device="cuda").manual_seed(9)
        for _ in range(3):
            tr.step(o, d, target, noise=torch.rand(2048, device="cuda", generator=g))
        assert tr.counters()["skipped"] == 0
        outs.append((tr.mlp_flat.clone(), tr.table.clone(), tr.mlp_m.clone()))
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        for x, y in zip(a, b):
            assert ((x - y).norm() / y.norm()).item() < 2e-4            # (summation order of the weight gradients; fp16 MLP)


def test_prefetched_march_placement_adapts_to_the_march_it_follows(hip_lib, lego_bitfield, monkeypatch):
    """Round 5: on one GPU with the optimizer in the scatter-add's flush the trainer places the next batch's march by the sample count
    of recent marches, reported asynchronously by the side stream (ngp_copy_to_host_async into ngp_host_alloc memory): a light march
    (trained-Lego occupancy: ~20 samples per ray) goes to the START of the step as 4-wave blocks, a heavy one (all cells occupied,
    ~500 samples per ray) before the scatter-add at low priority, and back when the grid thins out; with the result unchanged
    either way.  Setting one of the three NGP_EXPERIMENT keys (prefetch_at / march_shape / side_priority) pins the arrangement."""
    import ctypes
    from ngp_hip.trainer import FusedTrainer
    monkeypatch.delenv("NGP_EXPERIMENT", raising=False)
    n = 4096
    m, o, d, target = _make(lego_bitfield, n=n)
    tr = FusedTrainer(m, init_scale=2.0**10)
    assert tr._adaptive_prefetch and tr._hook_at == 3
    old_max = FusedTrainer._MARCH_NARROW_MAX
    sparse = m.density_bitfield.clone()
    try:
        FusedTrainer._MARCH_NARROW_MAX = 60 * n              # (the class threshold is in samples per step; scale it to this batch)
        hits0 = tr.prefetch_hits
        for i in range(6):
            tr.step(o, d, target, prefetch=(o, d))
            torch.cuda.synchronize()                         # the count of step i's prefetch has arrived before step i + 1 decides
        light = ctypes.c_int32.from_address(tr._marched_host.value).value
        assert 0 < light < 60 * n and tr._hook_at == 0 and tr.prefetch_hits - hits0 == 5
        m.density_bitfield.fill_(255)                        # every cell occupied: the marches get ~20 x heavier
        for i in range(4):
            tr.step(o, d, target, prefetch=(o, d))
            torch.cuda.synchronize()
        heavy = ctypes.c_int32.from_address(tr._marched_host.value).value
        assert heavy > 60 * n and tr._hook_at == 3
        m.density_bitfield.copy_(sparse)                     # the grid thins out: the counts say so, the next steps go back
        for i in range(3):
            tr.step(o, d, target, prefetch=(o, d))
            torch.cuda.synchronize()
        assert 0 < ctypes.c_int32.from_address(tr._marched_host.value).value < 60 * n and tr._hook_at == 0
    finally:
        FusedTrainer._MARCH_NARROW_MAX = old_max
    tr.close()
    assert tr._marched_host is None and tr._side_low is None
    monkeypatch.setenv("NGP_EXPERIMENT", "prefetch_at=2")
    tr2 = FusedTrainer(_make(lego_bitfield, n=n)[0])
    assert not tr2._adaptive_prefetch and tr2._prefetch_at == 2 and tr2._marched_host is None


def test_host_side_wait_for_the_prefetched_march_changes_nothing(hip_lib, lego_bitfield, monkeypatch):
    """Round 6, opt-in (NGP_EXPERIMENT prefetch_host_wait=1): where the next batch's march was issued at the START of the previous step, the
    HOST waits for its event (ngp_event_synchronize) instead of the main stream (a satisfied cross-queue event wait costs it ~10 us).  Same
    ordering guarantee, so: identical parameters after a run of prefetched steps, in deterministic mode, with either form of the wait."""
    from ngp_hip.trainer import FusedTrainer
    n = 4096

    def run(env):
        if env:
            monkeypatch.setenv("NGP_EXPERIMENT", env)
        else:
            monkeypatch.delenv("NGP_EXPERIMENT", raising=False)
        m, o, d, target = _make(lego_bitfield, n=n)
        tr = FusedTrainer(m, init_scale=2.0**10).set_deterministic(True)
        early = []
        for i in range(12):
            tr.step(o, d, target, prefetch=(o, d))
            torch.cuda.synchronize()                   # (the adaptive placement reads the previous march's count: make it arrive)
            early.append(tr._march_sets(n)[tr._cur].issued_early)
        tr.sync_master()
        out = (tr.table.clone(), tr.mlp_flat.clone(), tr.prefetch_hits, tr._host_wait_ok, early)
        tr.close()
        return out
    t_h, w_h, hits_h, ok_h, early_h = run("prefetch_host_wait=1")
    t_s, w_s, hits_s, ok_s, early_s = run("")
    assert ok_h and not ok_s and hits_h == hits_s == 11
    assert any(early_h) and early_h == early_s          # light marches go to the start of the step: that is where the host may wait
    assert torch.equal(t_h, t_s) and torch.equal(w_h, w_s)


def test_host_word_and_finite_check_entries(hip_lib):
    """The small entry points of round 5 on their own: pinned host memory + asynchronous device-to-host copy, and the read-only
    multi-tensor inf / nan check GradScaler's decision rests on."""
    import ctypes
    from ngp_hip import ops
    L = ops._lib()
    h = ctypes.c_void_p()
    assert L.ngp_host_alloc(ctypes.byref(h), 64) == 0 and h.value
    src = torch.tensor([123456789, 7], device="cuda", dtype=torch.int32)
    assert L.ngp_copy_to_host_async(h, ops._ptr(src), 8, ops._stream()) == 0
    torch.cuda.synchronize()
    assert (ctypes.c_int32 * 2).from_address(h.value)[:] == [123456789, 7]
    assert L.ngp_host_free(h) == 0
    assert L.ngp_host_alloc(None, 64) == -1 and L.ngp_copy_to_host_async(None, ops._ptr(src), 8, ops._stream()) == -1
    g = torch.Generator(device="cuda").manual_seed(1)
    ts = [torch.randn(n_, device="cuda", generator=g) for n_ in (4, 4096, 1 << 20, 12)]
    P = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    N = (ctypes.c_longlong * 4)(*[t.numel() for t in ts])
    found = torch.zeros(1, device="cuda")
    assert L.ngp_check_finite_multi(4, P, N, ops._ptr(found), ops._stream()) == 0
    assert float(found) == 0.0
    for which, pos, bad in ((0, 3, float("inf")), (2, (1 << 20) - 1, float("nan")), (3, 0, float("-inf")), (1, 2049, float("nan"))):
        keep = float(ts[which][pos])
        ts[which][pos] = bad
        found.zero_()
        assert L.ngp_check_finite_multi(4, P, N, ops._ptr(found), ops._stream()) == 0
        assert float(found) == 1.0, (which, pos)
        ts[which][pos] = keep
    found.zero_()
    assert L.ngp_check_finite_multi(4, P, N, ops._ptr(found), ops._stream()) == 0 and float(found) == 0.0
    N[3] = 13                                                  # not a multiple of four
    assert L.ngp_check_finite_multi(4, P, N, ops._ptr(found), ops._stream()) == -1
