"""Round 5 -- chunked forward: do not shade what compositing will never read.

The reference shades every sample raymarching_train emits and then ignores those behind T <= 1e-4 (volume_train.py:38; their `ws`
is not even initialised, :91-94); its own evaluation loop (rendering.py:62-158) shades in rounds and drops finished rays.
FusedTrainer(chunked_forward=True) shades a ray's samples in chunks of 64 / 64 / 128 / 256 / 512 and stops at the chunk in which the
ray's transmittance reaches the threshold.  Held here:
  * the list forms of the encoder and of the MLP forward are bit-identical, row by row, to the whole-buffer forms, and touch no other row;
  * the scheduler emits exactly the samples a float64 reference of its rule selects, round by round;
  * a trainer with the chunked forward holds bit-identical state (table, moments, MLP, GradScaler state) and per-ray outputs to one
    that shades everything, over steps that include grid updates, on the C3 shape (6 cascades, exponential stepping), with and without
    the distortion loss -- in deterministic mode, where nothing but the shaded set differs between the two;
  * what it shades lies between the composited and the marched sample counts (this young model is nearly transparent: almost
    everything is shaded; the saving on a trained scene is in profiles/r05_bench_garden_c3_march_placement.txt).
"""
import ctypes

import numpy as np
import pytest
import torch

from ngp_hip import ops

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int32) if t.dtype == torch.float32 else t.contiguous().view(torch.int16)


@pytest.mark.parametrize("pairs", [0, 1], ids=["natural", "pair-major"])
@pytest.mark.parametrize("kind", [0, 1], ids=["f32", "bf16copy"])
def test_list_encoder_and_mlp_equal_whole_buffer_forms(hip_lib, pairs, kind):
    L = ops._lib()
    lv = ops.make_levels(2**19, 16, 16, 4096, 2)
    g = torch.Generator(device="cuda").manual_seed(3)
    cap = 50000
    xyz = torch.rand(cap, 3, device="cuda", generator=g) * 2 - 1
    dirs = torch.randn(cap, 3, device="cuda", generator=g)
    table = (torch.rand(lv.total_entries * 2, device="cuda", generator=g) * 2 - 1) * 0.3
    tb = ops.cast_bf16(table) if kind else table
    ws = [torch.randn(s, device="cuda", generator=g) * 0.3 for s in ops.MLP_SHAPES]
    wpack = torch.empty(L.ngp_mlp_wpack_halfs(), device="cuda", dtype=torch.float16)
    assert L.ngp_mlp_pack(*[ops._ptr(w) for w in ws], pairs, ops._ptr(wpack), ops._stream()) == 0
    n_all = torch.tensor([cap], device="cuda", dtype=torch.int32)
    enc_ref = torch.empty(cap, 32, device="cuda")
    fwd = L.ngp_hash_fwd_bf16_ex if kind else L.ngp_hash_fwd_f32_ex
    assert fwd(ops._ptr(xyz), ops._ptr(tb), ctypes.byref(lv), cap, ops._ptr(n_all), 1, -1.0, 1.0, pairs, ops._ptr(enc_ref), ops._stream()) == 0
    sig_ref, rgb_ref = torch.empty(cap, device="cuda"), torch.empty(cap, 3, device="cuda", dtype=torch.float16)
    assert L.ngp_mlp_fwd_ex(ops._ptr(enc_ref), ops._ptr(dirs), ops._ptr(wpack), cap, ops._ptr(n_all), pairs, ops._ptr(sig_ref), ops._ptr(rgb_ref),
                            ops._stream()) == 0
    for n_list in (0, 1, 37, 4097, 33333):
        perm = torch.randperm(cap, device="cuda", generator=g)[:max(n_list, 1)].to(torch.int32).contiguous()
        cnt = torch.tensor([n_list], device="cuda", dtype=torch.int32)
        enc = torch.full((cap, 32), -7.0, device="cuda")
        assert L.ngp_hash_fwd_list(ops._ptr(xyz), ops._ptr(tb), kind, ctypes.byref(lv), cap, ops._ptr(cnt), ops._ptr(perm), 1, -1.0, 1.0, pairs,
                                   ops._ptr(enc), ops._stream()) == 0
        sig = torch.full((cap,), -7.0, device="cuda")
        rgb = torch.full((cap, 3), -7.0, device="cuda", dtype=torch.float16)
        enc_in = enc_ref.clone()                                     # the MLP list form on complete inputs, rows chosen by the list
        assert L.ngp_mlp_fwd_list(ops._ptr(enc_in), ops._ptr(dirs), ops._ptr(wpack), cap, ops._ptr(cnt), ops._ptr(perm), pairs, ops._ptr(sig),
                                  ops._ptr(rgb), ops._stream()) == 0
        torch.cuda.synchronize()
        rows = perm[:n_list].long()
        mask = torch.zeros(cap, dtype=torch.bool, device="cuda"); mask[rows] = True
        if pairs:      # pair-major planes [8][cap][4]: row r of plane p = floats (p * cap + r) * 4 ..
            e, er = enc.view(8, cap, 4), enc_ref.view(8, cap, 4)
            assert torch.equal(_bits(e[:, mask]), _bits(er[:, mask])) and bool((e[:, ~mask] == -7.0).all())
        else:
            assert torch.equal(_bits(enc[mask]), _bits(enc_ref[mask])) and bool((enc[~mask] == -7.0).all())
        assert torch.equal(_bits(sig[mask]), _bits(sig_ref[mask])) and bool((sig[~mask] == -7.0).all())
        assert torch.equal(_bits(rgb[mask]), _bits(rgb_ref[mask])) and bool((rgb[~mask] == -7.0).all())


def test_list_encoder_refuses_other_table_shapes(hip_lib):
    L = ops._lib()
    lv = ops.make_levels(2**19, 8, 16, 512, 2)
    x = torch.zeros(64, 3, device="cuda"); t = torch.zeros(lv.total_entries * 2, device="cuda"); out = torch.zeros(64, 16, device="cuda")
    cnt = torch.tensor([4], device="cuda", dtype=torch.int32); lst = torch.arange(4, device="cuda", dtype=torch.int32)
    assert L.ngp_hash_fwd_list(ops._ptr(x), ops._ptr(t), 0, ctypes.byref(lv), 64, ops._ptr(cnt), ops._ptr(lst), 0, 0.0, 1.0, 0, ops._ptr(out),
                               ops._stream()) == -2


def test_chunk_schedule_against_float64_rule(hip_lib):
    """Rays of 0 ... 700 samples in shuffled ranges; densities such that rays die in every round.  Round by round the emitted set is
    {start + j : begin <= j < min(begin + len, N), ray alive}, alive iff exp(-sum_{j < begin} sigma delta) > thr_stop (float64
    reference; rays within 1e-3 relative of the boundary excepted), each ray's range contiguous in the list."""
    L = ops._lib()
    rng = np.random.default_rng(11)
    n = 3000
    counts = rng.integers(0, 700, n).astype(np.int32)
    counts[:5] = [0, 1, 63, 64, 65]
    order = rng.permutation(n)
    starts = np.zeros(n, np.int64)
    starts[order] = np.concatenate([[0], np.cumsum(counts[order])[:-1]])
    total = int(counts.sum())
    rays_a = np.stack([np.arange(n), starts, counts], 1).astype(np.int32)
    sig = (rng.random(total) ** 4 * 3.0).astype(np.float32)
    kill = rng.random(n) < 0.5                                   # half the rays hit a wall somewhere
    wall = (rng.random(n) * np.maximum(counts, 1)).astype(np.int64)
    for r in np.nonzero(kill)[0]:
        if counts[r]:
            sig[starts[r] + wall[r]] = 8000.0                        # sigma delta >= 16: T falls below any threshold here
    dl = (rng.random(total) * 0.02 + 0.002).astype(np.float32)
    thr_stop = 0.5e-4
    d = lambda a: torch.from_numpy(a).cuda()
    ra, sg, de = d(rays_a), d(sig), d(dl)
    T_state = torch.full((n,), float("nan"), device="cuda")
    lst = torch.empty(total + 64, device="cuda", dtype=torch.int32)
    cnts = torch.zeros(2, 5, device="cuda", dtype=torch.int32)
    cnts[1] = 99
    rounds = [(0, 64, 0), (64, 64, 0), (128, 128, 64), (256, 256, 128), (512, 512, 256)]
    sd = sig.astype(np.float64) * dl.astype(np.float64)
    shaded_hi = np.zeros(n, np.int64)
    for r, (b, l, pb) in enumerate(rounds):
        assert L.ngp_chunk_schedule(ops._ptr(ra), ops._ptr(sg), ops._ptr(de), n, b, l, pb, thr_stop, ops._ptr(T_state), ops._ptr(lst),
                                    ops._ptr(cnts[0, r:r + 1]), ops._ptr(cnts[1, r:r + 1]), ops._stream()) == 0
        torch.cuda.synchronize()
        k = int(cnts[0, r])
        got = np.sort(lst[:k].cpu().numpy())
        want, near = [], 0
        for i in range(n):
            # alive at `begin` iff alive at every earlier round boundary (a retired ray stays retired)
            alive, close = True, False
            for (bb, _, _) in rounds[:r + 1]:
                if bb == 0:
                    continue
                T = np.exp(-sd[starts[i]:starts[i] + min(bb, counts[i])].sum())
                close |= abs(T - thr_stop) < 1e-3 * thr_stop
                alive &= T > thr_stop
            if close:
                near += 1
                continue
            if alive and b < counts[i]:
                want.append(np.arange(starts[i] + b, starts[i] + min(b + l, counts[i])))
                shaded_hi[i] = min(b + l, counts[i])
        want = np.sort(np.concatenate(want)) if want else np.zeros(0, np.int64)
        assert near < 5
        if near == 0:
            assert np.array_equal(got, want), r
        else:
            assert len(np.setdiff1d(want, got)) == 0 or len(np.setxor1d(want, got)) <= near * l
        assert int(cnts[1, r]) == 0                              # the other set's counter of this round was cleared
        # every ray's piece is contiguous in the list
        raw = lst[:k].cpu().numpy()
        brk = np.nonzero(np.diff(raw) != 1)[0]
        assert len(brk) + 1 <= n
    assert (shaded_hi <= counts).all() and shaded_hi.sum() < 0.9 * total          # the walls saved something
    # argument checks: boundaries off the 64 grid, a missing density buffer behind begin > 0
    assert L.ngp_chunk_schedule(ops._ptr(ra), ops._ptr(sg), ops._ptr(de), n, 32, 64, 0, thr_stop, ops._ptr(T_state), ops._ptr(lst),
                                ops._ptr(cnts[0, :1]), None, ops._stream()) == -1
    assert L.ngp_chunk_schedule(ops._ptr(ra), None, ops._ptr(de), n, 64, 64, 0, thr_stop, ops._ptr(T_state), ops._ptr(lst),
                                ops._ptr(cnts[0, :1]), None, ops._stream()) == -1


def _garden_trainer(chunked, w_dist, table_dtype=None, n=4096):
    from modules.networks import NGP
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    torch.manual_seed(0)
    m = NGP(scale=16.0, max_res=4096, table_dtype=table_dtype).cuda()
    m.density_bitfield.copy_(torch.from_numpy(synthetic.ball_slab_bitfield(6, 16.0, seed=7)).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    tr = FusedTrainer(m, lr=1e-2, max_steps=2000, exp_step_factor=1 / 256, distortion_loss_w=w_dist, init_scale=2.0**10,
                      chunked_forward=chunked)
    tr.set_deterministic(True)
    pool = []
    for b in range(3):
        o, d = synthetic.garden_rays(n, seed=40 + b)
        o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        pool.append((o, d, synthetic.garden_render_gt(o, d, scale=16.0).contiguous()))
    return m, tr, pool


@pytest.mark.parametrize("variant", ["f32", "f32+distortion", "bf16copy"])
def test_trainer_chunked_forward_is_bit_identical_to_shading_everything(hip_lib, variant):
    w = 1e-3 if variant.endswith("distortion") else 0.0
    td = torch.bfloat16 if variant == "bf16copy" else None
    res = []
    for chunked in (True, False):
        m, tr, pool = _garden_trainer(chunked, w, td)
        assert tr.chunked == chunked
        torch.manual_seed(5)
        outs, shaded = [], []
        for i in range(20):
            if i in (0, 8, 16):
                tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=i == 0)
            o, d, tgt = pool[i % 3]
            out = tr.step(o, d, tgt)
            if i in (3, 19):
                outs.append({k: out[k].clone() for k in ("rgb", "opacity", "depth", "vr_per_ray")} | {"rm": out["rm_samples"].clone()})
                shaded.append((tr.shaded_samples(), int(out["rm_samples"][0]), int(out["vr_per_ray"].sum())))
        torch.cuda.synchronize()
        assert tr.counters()["skipped"] == 0
        state = {"table": tr.table.clone(), "m": tr.table_m.clone(), "v": tr.table_v.clone(), "mlp": tr.mlp_flat.clone(),
                 "mlp_m": tr.mlp_m.clone(), "sf": tr.state_f.clone(), "si": tr.state_i.clone(), "grid": m.density_grid.clone(),
                 "bits": m.density_bitfield.clone()}
        res.append((state, outs, shaded))
    (sa, oa, sha), (sb, ob, shb) = res
    for k in sa:
        assert torch.equal(sa[k].view(torch.uint8), sb[k].view(torch.uint8)), k
    for a, b in zip(oa, ob):
        for k in a:
            assert torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)), k
    assert float(sa["table"].abs().max()) > 0
    for (sh, rm, vr), (sh0, rm0, vr0) in zip(sha, shb):
        assert sh0 is None and rm == rm0 and vr == vr0
        assert vr <= sh <= rm                                   # everything composited was shaded, nothing unmarched was


def test_chunked_forward_defaults(hip_lib):
    """On for multi-cascade / exponentially stepped scenes, off for the bounded synthetic ones; the half2 encoder shades everything."""
    from modules.networks import NGP
    from ngp_hip.trainer import FusedTrainer
    assert not FusedTrainer(NGP(scale=0.5, max_res=1024).cuda()).chunked
    assert FusedTrainer(NGP(scale=16.0, max_res=4096).cuda(), exp_step_factor=1 / 256).chunked
    assert [r[:2] for r in FusedTrainer(NGP(scale=16.0, max_res=4096).cuda(), exp_step_factor=1 / 256)._chunk_rounds] == \
        [(0, 64), (64, 64), (128, 128), (256, 256), (512, 512)]
    assert not FusedTrainer(NGP(scale=0.5, max_res=1024).cuda(), chunked_forward=True, max_samples=96).chunked


def test_live_list_and_per_ray_mse_gradient_match_the_one_block_forms(hip_lib):
    """Round 5, C3-sized batches: ngp_live_list (block-completion order, one atomic per 64 rays) holds the same samples as
    ngp_live_compact's ray-ordered list; ngp_mse_loss_grad_rays gives bit-identical gradients to the one-block ngp_mse_loss_grad and a
    per-ray squared error that sums to its loss."""
    L = ops._lib()
    rng = np.random.default_rng(4)
    n = 70000 - 3
    counts = rng.integers(0, 90, n).astype(np.int32)
    counts[:3] = [0, 1, 64]
    order = rng.permutation(n)
    starts = np.zeros(n, np.int64)
    starts[order] = np.concatenate([[0], np.cumsum(counts[order])[:-1]])
    rays_a = np.stack([np.arange(n), starts, counts], 1).astype(np.int32)
    perm_rows = rng.permutation(n)                                # rays_a rows in any order (the fused march's block-completion order)
    ra = torch.from_numpy(rays_a[perm_rows]).cuda()
    vr = torch.from_numpy((counts * rng.random(n)).astype(np.int32)).cuda()          # by RAY index
    total = int(counts.sum())
    a_idx, b_idx = torch.full((total + 64,), -1, device="cuda", dtype=torch.int32), torch.full((total + 64,), -1, device="cuda", dtype=torch.int32)
    a_tot, b_tot = torch.zeros(1, device="cuda", dtype=torch.int32), torch.zeros(1, device="cuda", dtype=torch.int32)
    other = torch.full((1,), 9, device="cuda", dtype=torch.int32)
    off = torch.empty(n, device="cuda", dtype=torch.int32)
    assert L.ngp_live_compact(ops._ptr(ra), ops._ptr(vr), n, ops._ptr(off), ops._ptr(a_idx), ops._ptr(a_tot), ops._stream()) == 0
    assert L.ngp_live_list(ops._ptr(ra), ops._ptr(vr), n, ops._ptr(b_idx), ops._ptr(b_tot), ops._ptr(other), ops._stream()) == 0
    torch.cuda.synchronize()
    k = int(a_tot)
    assert k == int(b_tot) == int(vr.sum()) and int(other) == 0
    assert torch.equal(torch.sort(a_idx[:k]).values, torch.sort(b_idx[:k]).values) and int(b_idx[k]) == -1
    # every ray's samples contiguous and ascending in the block-ordered list
    raw = b_idx[:k].cpu().numpy()
    assert (np.diff(raw) == 1).sum() >= k - n

    g = torch.Generator(device="cuda").manual_seed(2)
    rgb, op, tgt = torch.rand(n, 3, device="cuda", generator=g), torch.rand(n, device="cuda", generator=g), torch.rand(n, 3, device="cuda", generator=g)
    sf = torch.zeros(8, device="cuda"); sf[0] = 2.0**14
    for bg in (0.0, 1.0):
        g1, o1 = torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda")
        g2, o2, se = torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        sf1 = sf.clone()
        assert L.ngp_mse_loss_grad(ops._ptr(rgb), ops._ptr(op), ops._ptr(tgt), bg, n, ops._ptr(sf1), ops._ptr(g1), ops._ptr(o1), ops._stream()) == 0
        assert L.ngp_mse_loss_grad_rays(ops._ptr(rgb), ops._ptr(op), ops._ptr(tgt), bg, n, ops._ptr(sf), ops._ptr(g2), ops._ptr(o2), ops._ptr(se),
                                        ops._stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(_bits(g1), _bits(g2)) and torch.equal(_bits(o1), _bits(o2))
        assert abs(float(se.double().sum()) / (3 * n) - float(sf1[5])) < 1e-6 * float(sf1[5])
