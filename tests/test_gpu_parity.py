"""HIP (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars: bit-exact for everything integer / indexing / orbit (ray-AABB, march counts and samples, Morton, packbits,
hash corner selection via exact forward equality); tolerance-checked for reductions whose order differs
(compositing wave-scan: 1e-5 rel; hash backward atomics: 1e-5 rel)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ngp_hip import ops, synthetic  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def support_matches(ref, got, frac=1e-5):
    """Same set of touched entries, up to sums that cancel / underflow to exactly zero on one side."""
    bad = (ref != 0) != (got != 0)
    if bad.mean() > frac:
        return False
    scale = np.abs(ref).max()
    return bool(np.all(np.abs(ref[bad]) <= 1e-3 * scale) and np.all(np.abs(got[bad]) <= 1e-3 * scale))


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.fixture(scope="module")
def lego_batch(lego_bitfield, hip_lib):
    o, d = synthetic.lego_rays(8192, seed=23)
    rng = np.random.default_rng(5)
    noise = rng.random(8192, dtype=np.float32)
    return o, d, noise, lego_bitfield


def test_ray_aabb_bit_exact(oracle, lego_batch):
    o, d, _, _ = lego_batch
    # add misses, axis-parallel and inside-the-box rays
    o = o.copy(); d = d.copy()
    d[:64] = -d[:64]
    d[64:96, 0] = 0.0
    o[96:128] = 0.1
    for scale in (0.5, 16.0):
        ref = oracle.ray_aabb(o, d, scale)
        got = ops.ray_aabb(dev(o), dev(d), scale).cpu().numpy()
        assert bits_equal(ref, got)


@pytest.mark.parametrize("regime", ["lego", "random50", "ones"])
def test_march_train_bit_exact(oracle, lego_batch, regime):
    o, d, noise, bits = lego_batch
    n = 2048 if regime != "lego" else 8192
    o, d, noise = o[:n], d[:n], noise[:n]
    if regime == "random50":
        bits = synthetic.random_bitfield(1, fraction=0.5, seed=3)
    elif regime == "ones":
        bits = np.full(128**3 // 8, 255, np.uint8)
    hits = oracle.ray_aabb(o, d, 0.5)
    ra, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
    g_ra, g_x, g_d, g_dl, g_t, g_total = ops.march_train(dev(o), dev(d), dev(hits), dev(bits), dev(noise), 1, 0.5, 0.0, 128, 1024)
    assert int(g_total) == total
    assert np.array_equal(ra, g_ra.cpu().numpy())
    assert bits_equal(ts, g_t.cpu().numpy())
    assert bits_equal(deltas, g_dl.cpu().numpy())
    assert bits_equal(xyzs, g_x.cpu().numpy())
    assert bits_equal(dirs, g_d.cpu().numpy())
    if regime == "lego":
        assert 10 * n < total < 40 * n          # SURVEY probe: ~19.8 samples/ray on the trained-Lego grid


def test_march_train_cascades_exp_step(oracle, hip_lib):
    """Garden shape: 6 cascades, exponential stepping, max_samples truncation."""
    o, d = synthetic.garden_rays(4096, seed=7)
    bits = synthetic.ball_slab_bitfield(6, 16.0, seed=7)
    noise = np.random.default_rng(1).random(4096, dtype=np.float32)
    hits = oracle.ray_aabb(o, d, 16.0)
    for max_samples in (1024, 37):
        ra, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, noise, 6, 16.0, 1 / 256, 128, max_samples)
        g = ops.march_train(dev(o), dev(d), dev(hits), dev(bits), dev(noise), 6, 16.0, 1 / 256, 128, max_samples)
        assert int(g[5]) == total and total > 0
        assert np.array_equal(ra, g[0].cpu().numpy())
        assert bits_equal(ts, g[4].cpu().numpy()) and bits_equal(deltas, g[3].cpu().numpy())
        assert bits_equal(xyzs, g[1].cpu().numpy())


@pytest.mark.parametrize("group", [16, 32, 64])
def test_march_train_replay_stress(oracle, lego_batch, group, monkeypatch):
    """The count kernel resolves a batch of G orbit points in parallel (chain of examined points by pointer doubling): hold it to
    the serial oracle on sparse .. nearly full grids (short and long skips, chains of every length), with max_samples reached in
    the middle of a batch, for every lanes-per-ray variant."""
    monkeypatch.setenv("NGP_EXPERIMENT", "march_group=%d" % group)
    o, d, noise, _ = lego_batch
    n = 1024
    o, d, noise = o[:n].copy(), d[:n].copy(), noise[:n]
    hits = oracle.ray_aabb(o, d, 0.5)
    for fraction in (0.02, 0.3, 0.97):
        bits = synthetic.random_bitfield(1, fraction=fraction, seed=int(fraction * 100))
        for max_samples in (1, 7, 64, 1024):
            ref = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, max_samples)
            g = ops.march_train(dev(o), dev(d), dev(hits), dev(bits), dev(noise), 1, 0.5, 0.0, 128, max_samples)
            assert int(g[5]) == ref[5], (fraction, max_samples)
            assert np.array_equal(ref[0], g[0].cpu().numpy())
            assert bits_equal(ref[4], g[4].cpu().numpy()) and bits_equal(ref[3], g[3].cpu().numpy())
            assert bits_equal(ref[1], g[1].cpu().numpy())
    # two cascades, exponential stepping, coarse cells: skips that span several batches
    o2, d2 = synthetic.garden_rays(n, seed=11)
    hits2 = oracle.ray_aabb(o2, d2, 1.0)
    for fraction in (0.05, 0.6):
        bits = synthetic.random_bitfield(2, fraction=fraction, seed=9)
        for max_samples in (5, 256):
            ref = oracle.march_train(o2, d2, hits2, bits, noise, 2, 1.0, 1 / 256, 128, max_samples)
            g = ops.march_train(dev(o2), dev(d2), dev(hits2), dev(bits), dev(noise), 2, 1.0, 1 / 256, 128, max_samples)
            assert int(g[5]) == ref[5], (fraction, max_samples)
            assert np.array_equal(ref[0], g[0].cpu().numpy())
            assert bits_equal(ref[4], g[4].cpu().numpy()) and bits_equal(ref[3], g[3].cpu().numpy())
    # Garden shape (six cascades, cells up to 64 steps long): one skip covers several batches
    o3, d3 = synthetic.garden_rays(n, seed=7)
    bits = synthetic.ball_slab_bitfield(6, 16.0, seed=7)
    hits3 = oracle.ray_aabb(o3, d3, 16.0)
    for max_samples in (37, 1024):
        ref = oracle.march_train(o3, d3, hits3, bits, noise, 6, 16.0, 1 / 256, 128, max_samples)
        g = ops.march_train(dev(o3), dev(d3), dev(hits3), dev(bits), dev(noise), 6, 16.0, 1 / 256, 128, max_samples)
        assert int(g[5]) == ref[5] and ref[5] > 0
        assert np.array_equal(ref[0], g[0].cpu().numpy())
        assert bits_equal(ref[4], g[4].cpu().numpy()) and bits_equal(ref[3], g[3].cpu().numpy())


def _check_fused_march(ref, g, n):
    """ngp_march_train_fused vs the oracle's ray-order packing: per ray the same samples bit for bit; the rays' ranges tile
    [0, total) in some order (the reference's own order is whatever its atomic adds produce, ray_march.py:76-80)."""
    ra, xyzs, dirs, deltas, ts, total = ref
    g_ra, g_x, g_d, g_dl, g_t, g_total, ctr = [x.cpu().numpy() for x in g]
    assert int(g_total) == total and ctr.tolist() == [0, 0]                 # the counters are left zero for the next launch
    assert np.array_equal(g_ra[:, 0], np.arange(n)) and np.array_equal(g_ra[:, 2], ra[:, 2])
    order = np.argsort(g_ra[:, 1], kind="stable")
    nz = order[g_ra[order, 2] > 0]
    assert np.array_equal(g_ra[nz, 1], np.concatenate([[0], np.cumsum(g_ra[nz, 2])[:-1]]))          # no gap, no overlap
    # gather the fused output back into ray order and compare whole arrays
    idx = np.concatenate([np.arange(s, s + c) for _, s, c in g_ra] + [np.zeros(0, np.int64)]).astype(np.int64)
    assert len(idx) == total
    assert bits_equal(ts, g_t[idx]) and bits_equal(deltas, g_dl[idx]) and bits_equal(xyzs, g_x[idx]) and bits_equal(dirs, g_d[idx])


@pytest.mark.parametrize("regime", ["lego", "random50", "ones"])
def test_march_train_fused_bit_exact_per_ray(oracle, lego_batch, regime):
    o, d, noise, bits = lego_batch
    n = 2048 if regime != "lego" else 8192
    o, d, noise = o[:n], d[:n], noise[:n]
    if regime == "random50":
        bits = synthetic.random_bitfield(1, fraction=0.5, seed=3)
    elif regime == "ones":
        bits = np.full(128**3 // 8, 255, np.uint8)
    hits = oracle.ray_aabb(o, d, 0.5)
    ref = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
    for h in (dev(hits), None):                                              # explicit hits_t and the inline slab test
        g = ops.march_train_fused(dev(o), dev(d), h, dev(bits), dev(noise), 1, 0.5, 0.0, 128, 1024, capacity=ref[5] + 64)
        _check_fused_march(ref, g, n)


def test_march_train_fused_cascades_exp_step_ragged(oracle, hip_lib):
    """Garden shape (6 cascades, exponential stepping, truncation at max_samples) on a ray count that fills the last block partly."""
    n = 4096 - 37
    o, d = synthetic.garden_rays(4096, seed=7)
    o, d = o[:n], d[:n]
    bits = synthetic.ball_slab_bitfield(6, 16.0, seed=7)
    noise = np.random.default_rng(1).random(n, dtype=np.float32)
    hits = oracle.ray_aabb(o, d, 16.0)
    for max_samples in (1024, 37):
        ref = oracle.march_train(o, d, hits, bits, noise, 6, 16.0, 1 / 256, 128, max_samples)
        g = ops.march_train_fused(dev(o), dev(d), dev(hits), dev(bits), dev(noise), 6, 16.0, 1 / 256, 128, max_samples)
        _check_fused_march(ref, g, n)
    # a second launch on the same (self-resetting) counters: reuse ctr through the C entry directly is what the trainer does;
    # here: two launches in a row give the same per-ray result
    g2 = ops.march_train_fused(dev(o), dev(d), dev(hits), dev(bits), dev(noise), 6, 16.0, 1 / 256, 128, 37)
    _check_fused_march(ref, g2, n)


@pytest.mark.parametrize("shape", [(4, 0), (4, 81 * 1024), (8, 40 * 1024), (16, 0)])
def test_march_train_fused_shaped_bit_exact_per_ray(oracle, lego_batch, shape):
    """Round 5: ngp_march_train_fused_shaped -- 4- / 8- / 16-wave blocks, with and without the idle LDS that caps the blocks per CU:
    per ray the oracle's samples bit for bit (explicit jitter vector), ragged ray count, and the in-kernel jitter form against the
    vector form of the same seed."""
    o, d, noise, bits = lego_batch
    n = 4096 - 13
    o, d, noise = o[:n], d[:n], noise[:n]
    hits = oracle.ray_aabb(o, d, 0.5)
    ref = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
    g = ops.march_train_fused(dev(o), dev(d), None, dev(bits), dev(noise), 1, 0.5, 0.0, 128, 1024, shape=shape)
    _check_fused_march(ref, g, n)
    u = ops.rng_uniform(77, n).cpu().numpy()
    ref = oracle.march_train(o, d, hits, bits, u, 1, 0.5, 0.0, 128, 1024)
    g = ops.march_train_fused(dev(o), dev(d), dev(hits), dev(bits), None, 1, 0.5, 0.0, 128, 1024, seed=77, shape=shape)
    _check_fused_march(ref, g, n)
    # six cascades + exponential stepping + truncation through the same shapes
    o3, d3 = synthetic.garden_rays(1024, seed=5)
    bits3 = synthetic.ball_slab_bitfield(6, 16.0, seed=7)
    nz = np.random.default_rng(2).random(1024, dtype=np.float32)
    hits3 = oracle.ray_aabb(o3, d3, 16.0)
    ref = oracle.march_train(o3, d3, hits3, bits3, nz, 6, 16.0, 1 / 256, 128, 37)
    g = ops.march_train_fused(dev(o3), dev(d3), dev(hits3), dev(bits3), dev(nz), 6, 16.0, 1 / 256, 128, 37, shape=shape)
    _check_fused_march(ref, g, 1024)


def test_march_train_fused_shaped_rejects_bad_shapes(hip_lib, lego_batch):
    o, d, noise, bits = lego_batch
    for shape in ((3, 0), (32, 0), (4, -1), (4, 200 * 1024)):
        with pytest.raises(RuntimeError):
            ops.march_train_fused(dev(o[:64]), dev(d[:64]), None, dev(bits), dev(noise[:64]), 1, 0.5, 0.0, 128, 64, shape=shape)


def test_march_test_bit_exact(oracle, lego_batch):
    o, d, _, bits = lego_batch
    o, d = o[:4096], d[:4096]
    hits = oracle.ray_aabb(o, d, 0.5)
    alive = np.arange(0, 4096, 3, dtype=np.int64)
    h_ref = hits.copy()
    h_gpu = dev(hits)
    for n_step in (1, 4, 64):     # three consecutive resumed rounds
        r_idx, valid, deltas, ts, cnt = oracle.march_test(o, d, h_ref, alive, bits, 1, 0.5, 0.0, 128, n_step)
        g = ops.march_test(dev(o), dev(d), h_gpu, dev(alive), dev(bits), 1, 0.5, 0.0, 128, n_step)
        m = valid.astype(bool)
        assert np.array_equal(valid, g[1].cpu().numpy())
        assert np.array_equal(cnt, g[4].cpu().numpy())
        assert np.array_equal(r_idx[m], g[0].cpu().numpy()[m])
        assert bits_equal(ts[m], g[3].cpu().numpy()[m]) and bits_equal(deltas[m], g[2].cpu().numpy()[m])
        assert bits_equal(h_ref, h_gpu.cpu().numpy())


@pytest.mark.parametrize("max_res,features,levels", [(1024, 2, 16), (4096, 2, 16), (128, 4, 4)])
def test_hash_fwd_f32(oracle, hip_lib, max_res, features, levels):
    log2_T = 19 if features == 2 else 21
    base = 16 if features == 2 else 32
    lv = oracle.make_levels(2**log2_T, levels, base, max_res, features)
    lv_hip = ops.make_levels(2**log2_T, levels, base, max_res, features)
    assert bytes(lv) == bytes(lv_hip)                      # both sides index with the same table
    rng = np.random.default_rng(0)
    n = 20000
    x = rng.random((n, 3), dtype=np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.25, 0.75, 1.0], [0.999999, 1e-7, 0.5]]
    table = rng.random(lv.total_entries * features, dtype=np.float32)
    ref = oracle.hash_fwd_f32(x, table, lv)
    got = ops.hash_fwd_f32(dev(x), dev(table), lv_hip).cpu().numpy()
    # same corners, same weights, same add order => bit-exact
    assert bits_equal(ref, got)


@pytest.mark.parametrize("max_params", [3 * 2**17, 2**19 - 8, 100000])
def test_hash_fwd_f32_real_modulo_levels(oracle, hip_lib, max_params):
    """Tables whose hashed levels are NOT a power of two entries (index = hash % size, hash_encoder.py:71; no shipped config has
    one): the round-4 gather loop takes the `%` under its one rarely taken branch (hash_grid.hip corners_flat), per lane -- the
    dense levels of the same launch keep the conditional subtract.  Bit-exact against the oracle, natural and pair-major layout."""
    lv = ops.make_levels(max_params, 16, 16, 1024, 2)
    assert bytes(lv) == bytes(oracle.make_levels(max_params, 16, 16, 1024, 2))
    sizes = [int(v) for v in np.ctypeslib.as_array(lv.map_size)[:16]]
    assert any(v & (v - 1) for v in sizes[int(lv.begin_fast_hash_level):])       # at least one hashed level needs the real modulo
    rng = np.random.default_rng(4)
    n = 20000
    x = rng.random((n, 3), dtype=np.float32)
    x[:6] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0.25, 0.75, 1.0], [0.999999, 1e-7, 0.5]]
    table = rng.random(lv.total_entries * 2, dtype=np.float32)
    ref = oracle.hash_fwd_f32(x, table, lv)
    assert bits_equal(ref, ops.hash_fwd_f32(dev(x), dev(table), lv).cpu().numpy())
    L = ops._lib()
    xd, td = dev(x), dev(table)
    pm = torch.empty(8 * n * 4, device="cuda")
    assert L.ngp_hash_fwd_f32_ex(ops._ptr(xd), ops._ptr(td), ctypes.byref(lv), n, ops._ptr(None), 0, 0.0, 1.0, 1, ops._ptr(pm),
                                 ops._stream()) == 0
    got = pm.view(8, n, 2, 2).cpu().numpy()                # [pair][sample][level p | level 15 - p][feature]
    nat = ref.reshape(n, 16, 2)
    for pr in range(8):
        assert bits_equal(got[pr, :, 0], nat[:, pr]) and bits_equal(got[pr, :, 1], nat[:, 15 - pr])


@pytest.mark.parametrize("lo,hi", [(-0.5, 0.5), (-0.3, 0.3), (-8.0, 8.0), (-1.7, 2.9)])
def test_hash_fwd_f32_fused_normalisation(oracle, hip_lib, lo, hi):
    """The gather's fused (x - lo) / (hi - lo) (networks.py:144): one multiply where hi - lo is a power of two, the IEEE division
    elsewhere -- both must equal numpy's float32 subtract-then-divide bit for bit, then the oracle's encoding of that."""
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(5)
    n = 8192
    x = (rng.random((n, 3), dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)
    x01 = ((x - np.float32(lo)) / (np.float32(hi) - np.float32(lo))).astype(np.float32)
    x01 = np.clip(x01, 0.0, 1.0)                           # (random points: rounding may leave [0, 1] by an ulp; keep both sides equal)
    keep = ((x - np.float32(lo)) / (np.float32(hi) - np.float32(lo)) == x01).all(axis=1)
    table = rng.random(lv.total_entries * 2, dtype=np.float32)
    ref = oracle.hash_fwd_f32(x01, table, lv)
    L = ops._lib()
    xd, td = dev(x), dev(table)
    out = torch.empty(n, 32, device="cuda")
    assert L.ngp_hash_fwd_f32_ex(ops._ptr(xd), ops._ptr(td), ctypes.byref(lv), n, ops._ptr(None), 1, lo, hi, 0, ops._ptr(out),
                                 ops._stream()) == 0
    assert keep.sum() > n * 0.99 and bits_equal(ref[keep], out.cpu().numpy()[keep])


def test_hash_bwd_f32(oracle, hip_lib):
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(1)
    n = 30000
    x = rng.random((n, 3), dtype=np.float32)
    dout = rng.standard_normal((n, 32)).astype(np.float32)
    dout[::7] = 0.0                                        # zero-gradient samples are skipped
    ref = oracle.hash_bwd_f32(x, dout, lv)
    dt = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32(dev(x), dev(dout), lv, dt)
    got = dt.cpu().numpy()
    assert support_matches(ref, got)                       # identical support = identical indexing
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("max_res", [1024, 4096])
def test_hash_bwd_f32_sliced(oracle, hip_lib, max_res):
    """The LDS-sliced scatter-add (no global float atomics; csrc/hash_bwd_lds.hip) against the oracle: same touched entries,
    2e-5; ray-like point runs exercise the equal-cell pre-summing and the per-wave hit queue; plus the live-list / device-count
    form and the accumulate-into semantics."""
    lv = ops.make_levels(2**19, 16, 16, max_res, 2)
    rng = np.random.default_rng(1)
    n_rays, per_ray = 600, 50
    o = rng.random((n_rays, 1, 3), dtype=np.float32) * 0.8 + 0.1
    d = rng.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(per_ray, dtype=np.float32) * np.float32(0.0017))[None, :, None]
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)             # consecutive samples share coarse cells
    x = np.concatenate([x, rng.random((5003, 3), dtype=np.float32),
                        np.array([[0, 0, 0], [1, 1, 1], [1, 0, 1], [0.999999, 1e-7, 0.5]], np.float32)])
    n = x.shape[0]
    dout = rng.standard_normal((n, 32)).astype(np.float32)
    dout[::7] = 0.0
    ref = oracle.hash_bwd_f32(x, dout, lv)
    dt = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32_sliced(dev(x), dev(dout), lv, dt)
    got = dt.cpu().numpy()
    assert support_matches(ref, got)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
    # accumulates into what is already there, like ngp_hash_bwd_f32
    ops.hash_bwd_f32_sliced(dev(x), dev(dout), lv, dt)
    np.testing.assert_allclose(dt.cpu().numpy(), 2 * ref, rtol=4e-5, atol=4e-5)
    # live list + device-side count: position j of dout belongs to sample live_idx[j]; only the first n_dev entries count
    perm = rng.permutation(n).astype(np.int32)
    m = n - 1234
    ref_live = oracle.hash_bwd_f32(x[perm[:m]], dout[:m], lv)
    dt2 = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32_sliced(dev(x), dev(dout), lv, dt2, live_idx=dev(perm), n_dev=torch.tensor([m], device="cuda", dtype=torch.int32))
    assert support_matches(ref_live, dt2.cpu().numpy())
    np.testing.assert_allclose(dt2.cpu().numpy(), ref_live, rtol=2e-5, atol=2e-5)
    # and it agrees with the float-atomic kernel
    dt3 = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32(dev(x), dev(dout), lv, dt3)
    np.testing.assert_allclose(got, dt3.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_hash_bwd_f32_sliced_concentrated_plan(oracle, hip_lib):
    """Round 5: the concentrated plan (NGP_BWD_PLAN_CONCENTRATED in the level table) -- coarse hashed levels with sample-range replicas -- on the C3
    table with points crowded into 2 % of the box (what a multi-cascade scene looks like to the coarse levels): same touched entries
    and values as the oracle, as the default plan, and through the optimizer-in-the-flush entry's level split."""
    L = ops._lib()
    lv = ops.make_levels(2**19, 16, 16, 4096, 2)
    rng = np.random.default_rng(3)
    n_rays, per_ray = 700, 60
    o = 0.5 + (rng.random((n_rays, 1, 3), dtype=np.float32) - 0.5) * 0.02
    d = rng.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(per_ray, dtype=np.float32) * np.float32(0.0003))[None, :, None]
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    x = np.concatenate([x, rng.random((3001, 3), dtype=np.float32)])
    dout = rng.standard_normal((x.shape[0], 32)).astype(np.float32)
    ref = oracle.hash_bwd_f32(x, dout, lv)
    nrep = (ctypes.c_uint8 * 16)()
    mm = ctypes.c_uint32()
    from ngp_hip import lib as _libmod
    lvc = lv.with_plan(_libmod.BWD_PLAN_CONCENTRATED)
    try:
        assert L.ngp_hash_bwd_sliced_plan(ctypes.byref(lvc), None, 0, None, None, nrep, ctypes.byref(mm), None) > 0
        assert nrep[5] > 1 and nrep[7] > 1 and nrep[8] == 1 and nrep[15] == 1  # hashed levels up to res 256: sample-range replicas
        prefix = int(L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lvc)))
        first = min(l for l in range(16) if nrep[l] == 1 and all(nrep[k] == 1 for k in range(l, 16)))
        assert prefix == lv.offset[first] * 2
        dt = torch.zeros(lv.total_entries * 2, device="cuda")
        ops.hash_bwd_f32_sliced(dev(x), dev(dout), lvc, dt)
        got = dt.cpu().numpy()
    finally:
        pass
    assert support_matches(ref, got)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * float(np.abs(ref).max()))
    dt0 = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32_sliced(dev(x), dev(dout), lv, dt0)
    np.testing.assert_allclose(got, dt0.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(np.abs(ref).max()))


@pytest.mark.parametrize("n", [1, 63, 65, 4097])
def test_hash_bwd_f32_sliced_ragged_sizes(oracle, hip_lib, n):
    """Sample counts that are not multiples of the 64-sample bitmap word / the 4096-sample super-chunk, down to one sample."""
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(100 + n)
    x = rng.random((n, 3), dtype=np.float32)
    dout = rng.standard_normal((n, 32)).astype(np.float32)
    ref = oracle.hash_bwd_f32(x, dout, lv)
    dt = torch.zeros(lv.total_entries * 2, device="cuda")
    ops.hash_bwd_f32_sliced(dev(x), dev(dout), lv, dt)
    got = dt.cpu().numpy()
    assert support_matches(ref, got)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


def test_hash_bwd_f32_sliced_empty_and_flags(oracle, hip_lib):
    """Device-side count 0 (every ray terminated at once / no ray hit the box): nothing is added, nothing hangs -- also right
    after a non-empty launch (the persistent workgroups' queue heads reset themselves).  A non-finite gradient raises the
    GradScaler flag, and only then."""
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(7)
    n = 3000
    x, dout = dev(rng.random((n, 3), dtype=np.float32)), dev(rng.standard_normal((n, 32)).astype(np.float32))
    L = ops._lib()
    ws = ops.sliced_workspace(lv, n, x.device)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)

    def run(count, d):
        dt = torch.zeros(lv.total_entries * 2, device="cuda")
        cnt = torch.tensor([count], device="cuda", dtype=torch.int32)
        rc = L.ngp_hash_bwd_f32_sliced(ops._ptr(x), ops._ptr(d), ctypes.byref(lv), n, ops._ptr(cnt), ops._ptr(None), 0, 0.0, 1.0, 0,
                                       ops._ptr(dt), ops._ptr(flag), ops._ptr(ws), ws.numel(), ops._stream())
        assert rc == 0
        return dt
    assert float(run(0, dout).abs().max()) == 0.0 and int(flag) == 0
    full = run(n, dout)
    assert float(full.abs().max()) > 0 and int(flag) == 0
    assert float(run(0, dout).abs().max()) == 0.0
    again = run(n, dout)
    np.testing.assert_allclose(again.cpu().numpy(), full.cpu().numpy(), rtol=1e-6, atol=1e-7)
    bad = dout.clone(); bad[1234, 7] = float("inf")
    run(n, bad)
    assert int(flag) == 1
    assert L.ngp_hash_bwd_f32_sliced(ops._ptr(x), ops._ptr(dout), ctypes.byref(lv), 0, ops._ptr(None), ops._ptr(None), 0, 0.0, 1.0, 0,
                                     ops._ptr(full), ops._ptr(None), ops._ptr(ws), ws.numel(), ops._stream()) == 0     # n_max = 0: no-op


def test_hash_bwd_operator_dispatch(oracle, hip_lib, monkeypatch):
    """ops.hash_bwd_f32 (what modules/hash_encoder.py's backward calls) takes the sliced form from SLICED_MIN_SAMPLES up and the
    float-atomic kernel below / when NGP_HASH_BWD=atomic / when the table does not fit: all three against the oracle."""
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(11)
    n = ops.SLICED_MIN_SAMPLES + 1000
    x = rng.random((n, 3), dtype=np.float32)
    dout = rng.standard_normal((n, 32)).astype(np.float32)
    ref = oracle.hash_bwd_f32(x, dout, lv)
    calls = []
    L = ops._lib()
    real = L.ngp_hash_bwd_f32_sliced
    monkeypatch.setattr(L, "ngp_hash_bwd_f32_sliced", lambda *a: (calls.append(1), real(*a))[1])
    got = ops.hash_bwd_f32(dev(x), dev(dout), lv, torch.zeros(lv.total_entries * 2, device="cuda")).cpu().numpy()
    assert calls == [1]
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
    monkeypatch.setenv("NGP_HASH_BWD", "atomic")
    got = ops.hash_bwd_f32(dev(x), dev(dout), lv, torch.zeros(lv.total_entries * 2, device="cuda")).cpu().numpy()
    assert calls == [1]
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
    monkeypatch.delenv("NGP_HASH_BWD")
    m = 5000                                                # a small batch stays on the atomic kernel
    got = ops.hash_bwd_f32(dev(x[:m]), dev(dout[:m]), lv, torch.zeros(lv.total_entries * 2, device="cuda")).cpu().numpy()
    assert calls == [1]
    np.testing.assert_allclose(got, oracle.hash_bwd_f32(x[:m], dout[:m], lv), rtol=2e-5, atol=2e-5)


def test_hash_bwd_f32_sliced_refuses_unsupported_tables(hip_lib):
    lv = ops.make_levels(2**21, 4, 32, 128, 4)             # F = 4: not expressible, the C entry point says so (-2)
    x = torch.rand(100, 3, device="cuda")
    with pytest.raises(RuntimeError, match="code -2"):
        ops.hash_bwd_f32_sliced(x, torch.zeros(100, 16, device="cuda"), lv, torch.zeros(lv.total_entries * 4, device="cuda"))


def test_hash_f16(oracle, hip_lib):
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(2)
    n = 20000
    x = rng.random((n, 3), dtype=np.float32)
    table = ((rng.random((lv.total_entries, 2), dtype=np.float32) * 2 - 1) * 1e-1).astype(np.float16)
    ref = oracle.hash_fwd_f16(x, table, lv)
    got = ops.hash_fwd_f16(dev(x), dev(table), lv).cpu().numpy()
    assert bits_equal(ref, got)                            # f16 accumulate in the same order: bit-exact
    dout = (rng.standard_normal((n, 16, 2)) * 1e-2).astype(np.float16)
    dout[::5] = 0
    ref_g = oracle.hash_bwd_f16(x, dout, lv)
    g = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(dev(x), dev(dout), lv, g)
    got_g = g.float().cpu().numpy()
    assert support_matches(ref_g, got_g)
    # f16 atomics round after every add: tolerance scales with the number of contributions on coarse levels
    np.testing.assert_allclose(got_g, ref_g, rtol=3e-2, atol=2e-3)


def test_hash_bwd_f16_sliced(oracle, hip_lib, monkeypatch):
    """The half2 encoder's scatter-add in the LDS-sliced form (ngp_hash_bwd_sliced_main_f16): same support as the oracle; it sums
    the fp16-rounded contributions exactly and rounds once, so it sits closer to the oracle (fp32 sum of the same contributions)
    than the packed-f16-atomic kernel can: one fp16 ulp of the result."""
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(5)
    n_rays, per_ray = 700, 48
    o = rng.random((n_rays, 1, 3), dtype=np.float32) * 0.8 + 0.1
    d = rng.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(per_ray, dtype=np.float32) * np.float32(0.0017))[None, :, None]
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    x = np.concatenate([x, rng.random((7001, 3), dtype=np.float32)])
    n = x.shape[0]
    dout = (rng.standard_normal((n, 16, 2)) * 1e-2).astype(np.float32)
    dout[::5] = 0
    ref = oracle.hash_bwd_f16(x, dout.astype(np.float16), lv)
    g = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16_sliced(dev(x), dev(dout.reshape(n, 32)), lv, g)
    got = g.float().cpu().numpy()
    assert support_matches(ref, got)
    # levels owned by ONE workgroup per slice (the hashed ones, 6..15 here): the exact sum, rounded once -- fp16's 2^-11 relative;
    # the absolute floor is one fp16 ulp of a single summand (|w g| ~ 4e-3..8e-3 -> 3.8e-6..7.6e-6: 1 entry in 1e7 sees one
    # contribution round the other way).  The coarse dense levels are replicated over sample ranges and their partial sums meet in packed fp16
    # atomics (<= 63 of them per entry, where the reference's formulation has one per contribution): the f16-atomic tolerance
    split = int(lv.offset[6])
    np.testing.assert_allclose(got[split:], ref[split:], rtol=1.5e-3, atol=8e-6)
    np.testing.assert_allclose(got[:split], ref[:split], rtol=3e-2, atol=2e-3)
    g_at = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(dev(x), dev(dout.astype(np.float16)), lv, g_at)
    err_sliced, err_atomic = np.abs(got - ref).max(), np.abs(g_at.float().cpu().numpy() - ref).max()
    assert err_sliced <= 1.5 * err_atomic, (err_sliced, err_atomic)      # never worse than one packed-f16 atomic per contribution
    # the live-list / device-count form
    perm = rng.permutation(n).astype(np.int32)
    m = n - 999
    ref_live = oracle.hash_bwd_f16(x[perm[:m]], dout[:m].astype(np.float16), lv)
    g2 = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16_sliced(dev(x), dev(dout.reshape(n, 32)), lv, g2, live_idx=dev(perm), n_dev=torch.tensor([m], device="cuda", dtype=torch.int32))
    got2 = g2.float().cpu().numpy()
    np.testing.assert_allclose(got2[split:], ref_live[split:], rtol=1.5e-3, atol=8e-6)
    np.testing.assert_allclose(got2[:split], ref_live[:split], rtol=3e-2, atol=2e-3)
    # the operator (modules/hash_encoder_half.py's backward) takes the same form for large batches
    monkeypatch.setattr(ops, "SLICED_MIN_SAMPLES", 1000)
    g3 = torch.zeros(lv.total_entries, 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(dev(x), dev(dout.astype(np.float16)), lv, g3)
    assert torch.equal(g3[split:], g[split:])                   # single-owner levels: deterministic
    np.testing.assert_allclose(g3[:split].float().cpu().numpy(), ref[:split], rtol=3e-2, atol=2e-3)    # (packed-f16 atomics: order)


def test_sh16(oracle, hip_lib):
    rng = np.random.default_rng(3)
    d = rng.random((10000, 3), dtype=np.float32)
    assert bits_equal(oracle.sh16_fwd(d), ops.sh16_fwd(dev(d)).cpu().numpy())
    g = rng.standard_normal((10000, 16)).astype(np.float32)
    np.testing.assert_allclose(ops.sh16_bwd(dev(d), dev(g)).cpu().numpy(), oracle.sh16_bwd(d, g), rtol=1e-5, atol=1e-5)


def _composite_case(rng, n_rays, max_len, dense=False):
    counts = rng.integers(0, max_len, n_rays).astype(np.int32)
    counts[::9] = 0
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    perm = rng.permutation(n_rays).astype(np.int32)          # ray_idx != row order
    rays_a = np.stack([perm, starts, counts], -1).astype(np.int32)
    S = int(counts.sum())
    sig = (rng.random(S, dtype=np.float32) * (60 if dense else 8)).astype(np.float32)
    rgbs = rng.random((S, 3), dtype=np.float32)
    deltas = np.full(S, 1.7320508 / 1024, np.float32)
    ts = (rng.random(S, dtype=np.float32) + 0.5).astype(np.float32)
    return rays_a, sig, rgbs, deltas, ts


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("dense", [False, True])
def test_composite_train_fwd_bwd(oracle, hip_lib, half, dense):
    rng = np.random.default_rng(4)
    rays_a, sig, rgbs, deltas, ts = _composite_case(rng, 3000, 300, dense)
    if half:
        rgbs = rgbs.astype(np.float16)
    rgbs32 = rgbs.astype(np.float32)
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(sig, rgbs32, deltas, ts, rays_a, 1e-4)
    g = ops.composite_train_fwd(dev(sig), dev(rgbs), dev(deltas), dev(ts), dev(rays_a), 1e-4)
    g_tot = g[0].cpu().numpy()
    # the early-termination boundary may move by one sample when T hovers at the threshold (different product order)
    assert np.abs(g_tot - tot).max() <= 1 and (g_tot != tot).mean() < 1e-2
    np.testing.assert_allclose(g[1].cpu().numpy(), op, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g[2].cpu().numpy(), dep, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g[3].cpu().numpy(), rgb, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g[4].cpu().numpy(), ws, rtol=1e-5, atol=2e-7)
    n = rays_a.shape[0]
    g_op = rng.standard_normal(n).astype(np.float32)
    g_dep = rng.standard_normal(n).astype(np.float32)
    g_rgb = rng.standard_normal((n, 3)).astype(np.float32)
    for g_ws in (None, rng.standard_normal(sig.shape[0]).astype(np.float32)):
        ds, dc = oracle.composite_train_bwd(g_op, g_dep, g_rgb, g_ws, sig, rgbs32, deltas, ts, rays_a, 1e-4)
        got_ds, got_dc = ops.composite_train_bwd(dev(g_op), dev(g_dep), dev(g_rgb), None if g_ws is None else dev(g_ws),
                                                 dev(sig), dev(rgbs), dev(deltas), dev(ts), dev(rays_a), g[1], g[2], g[3], g[4],
                                                 1e-4)
        scale = np.abs(ds).max()
        np.testing.assert_allclose(got_ds.cpu().numpy(), ds, rtol=1e-3, atol=2e-5 * scale)
        np.testing.assert_allclose(got_dc.float().cpu().numpy(), dc, rtol=2e-3 if half else 1e-4, atol=1e-6)


@pytest.mark.parametrize("bg", [1.0, 0.5])
def test_composite_train_background_blend_entries(oracle, hip_lib, bg):
    """Round 6: the blend of rendering.py:219-226 (rgb + bg (1 - opacity)) inside the compositing launches (ngp_composite_train_fwd_bg /
    _bwd_bg, what the fused render() calls).  Forward: the blended colour against the oracle's composite + the blend on the host, the
    unblended outputs bit-equal to the plain entry's.  Backward: against the oracle handed the blend's share of d opacity."""
    from ngp_hip.ops import _ptr, _stream
    rng = np.random.default_rng(14)
    rays_a, sig, rgbs, deltas, ts = _composite_case(rng, 2000, 200, True)
    rgbs16 = rgbs.astype(np.float16)
    n, S = rays_a.shape[0], sig.shape[0]
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(sig, rgbs16.astype(np.float32), deltas, ts, rays_a, 1e-4)
    plain = ops.composite_train_fwd(dev(sig), dev(rgbs16), dev(deltas), dev(ts), dev(rays_a), 1e-4)
    d_sig, d_rgbs, d_del, d_ts, d_ra = dev(sig), dev(rgbs16), dev(deltas), dev(ts), dev(rays_a)
    f32 = dict(device="cuda", dtype=torch.float32)
    g_tot = torch.empty(n, device="cuda", dtype=torch.int32)
    g_op, g_dep, g_rgb, g_out, g_ws = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(S, **f32)
    assert hip_lib.ngp_composite_train_fwd_bg(_ptr(d_sig), _ptr(d_rgbs), 1, _ptr(d_del), _ptr(d_ts), _ptr(d_ra), 1e-4, n, _ptr(g_tot), _ptr(g_op),
                                              _ptr(g_dep), _ptr(g_rgb), _ptr(g_ws), _ptr(g_out), bg, _stream()) == 0
    torch.cuda.synchronize()
    for a, b in zip(plain, (g_tot, g_op, g_dep, g_rgb, g_ws)):
        assert torch.equal(a, b)
    # (fp32 on both sides: one rounding of bg (1 - O), one of the sum; the host may contract differently by an ulp of the colour)
    np.testing.assert_allclose(g_out.cpu().numpy(), rgb + np.float32(bg) * (1 - op)[:, None], rtol=1e-5, atol=2e-6)
    assert torch.equal(g_out, g_rgb + bg * (1 - g_op)[:, None]) or bg != 1.0
    go = rng.standard_normal(n).astype(np.float32)
    gr = rng.standard_normal((n, 3)).astype(np.float32)
    ds, dc = oracle.composite_train_bwd(go - np.float32(bg) * gr.sum(1), np.zeros(n, np.float32), gr, None, sig, rgbs16.astype(np.float32), deltas, ts, rays_a, 1e-4)
    o_ds, o_dc = torch.empty(S, **f32), torch.empty(S, 3, device="cuda", dtype=torch.float16)
    assert hip_lib.ngp_composite_train_bwd_bg(_ptr(dev(go)), _ptr(None), _ptr(dev(gr)), _ptr(None), _ptr(d_sig), _ptr(d_rgbs), 1, _ptr(d_del),
                                              _ptr(d_ts), _ptr(d_ra), _ptr(g_op), _ptr(g_dep), _ptr(g_rgb), _ptr(g_ws), 1e-4, n, _ptr(o_ds),
                                              _ptr(o_dc), bg, _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(o_ds.cpu().numpy(), ds, rtol=1e-3, atol=2e-5 * np.abs(ds).max())
    np.testing.assert_allclose(o_dc.float().cpu().numpy(), dc, rtol=2e-3, atol=1e-6)


def test_composite_test_kernel(oracle, hip_lib):
    rng = np.random.default_rng(6)
    n_rays, n_alive = 5000, 1700
    alive = np.sort(rng.choice(n_rays, n_alive, replace=False)).astype(np.int64)
    counts = rng.integers(0, 9, n_alive).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    pack = np.stack([starts, counts], -1).astype(np.int64)
    S = int(counts.sum())
    sig = (rng.random(S, dtype=np.float32) * 500).astype(np.float32)
    rgbs = rng.random((S, 3), dtype=np.float32)
    deltas = np.full(S, 1.7320508 / 1024, np.float32)
    ts = rng.random(S, dtype=np.float32) + 0.5
    op = (rng.random(n_rays, dtype=np.float32) * 0.5).astype(np.float32); dep = rng.random(n_rays, dtype=np.float32)
    rgb = rng.random((n_rays, 3), dtype=np.float32)
    a_ref, op_ref, dep_ref, rgb_ref = alive.copy(), op.copy(), dep.copy(), rgb.copy()
    oracle.composite_test(sig, rgbs, deltas, ts, pack, a_ref, 1e-4, op_ref, dep_ref, rgb_ref)
    a_g, op_g, dep_g, rgb_g = dev(alive), dev(op), dev(dep), dev(rgb)
    ops.composite_test(dev(sig), dev(rgbs), dev(deltas), dev(ts), dev(pack), a_g, 1e-4, op_g, dep_g, rgb_g)
    assert np.array_equal(a_ref, a_g.cpu().numpy())
    np.testing.assert_allclose(op_g.cpu().numpy(), op_ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dep_g.cpu().numpy(), dep_ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rgb_g.cpu().numpy(), rgb_ref, rtol=1e-6, atol=1e-7)


def test_grid_utils_bit_exact(oracle, hip_lib):
    rng = np.random.default_rng(8)
    coords = rng.integers(0, 128, (100000, 3)).astype(np.int32)
    idx = oracle.morton3d(coords)
    assert np.array_equal(idx, ops.morton3d(dev(coords)).cpu().numpy())
    assert np.array_equal(oracle.morton3d_invert(idx), ops.morton3d_invert(dev(idx)).cpu().numpy())
    assert np.array_equal(coords, ops.morton3d_invert(dev(idx)).cpu().numpy())
    grid = rng.standard_normal(128**3).astype(np.float32)
    out = torch.zeros(128**3 // 8, dtype=torch.uint8, device="cuda")
    ops.packbits(dev(grid), 0.3, out)
    assert np.array_equal(oracle.packbits(grid, 0.3), out.cpu().numpy())


def test_full_size_properties(hip_lib, lego_bitfield):
    """BASELINE sizes (65536 rays, init-regime occupancy): size-independent invariants instead of an oracle run."""
    o, d = synthetic.lego_rays(65536, seed=11)
    bits = synthetic.random_bitfield(1, fraction=0.5, seed=4)
    noise = torch.rand(65536, device="cuda")
    hits = ops.ray_aabb(dev(o), dev(d), 0.5)
    rays_a, xyzs, dirs, deltas, ts, total = ops.march_train(dev(o), dev(d), hits, dev(bits), noise, 1, 0.5, 0.0, 128, 1024)
    ra = rays_a.cpu().numpy()
    assert np.array_equal(ra[:, 0], np.arange(65536))
    assert np.array_equal(ra[:, 1], np.concatenate([[0], np.cumsum(ra[:, 2])[:-1]]))      # exclusive scan
    assert int(total) == ra[:, 2].sum() == xyzs.shape[0] and ra[:, 2].max() <= 1024
    # samples of a ray are strictly increasing in t and lie inside [t1, t2)
    t = ts.cpu().numpy(); h = hits.cpu().numpy()
    seg = np.repeat(np.arange(65536), ra[:, 2])
    assert np.all(t >= h[seg, 0]) and np.all(t < h[seg, 1])
    same = seg[1:] == seg[:-1]
    assert np.all(np.diff(t)[same] > 0)
    assert float(xyzs.abs().max()) <= 0.5 + 1e-5
    # linearity of the encoder in the table and of its transpose
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    x01 = ((xyzs[:200000] + 0.5)).clamp(0, 1).contiguous()
    t1 = torch.rand(lv.total_entries * 2, device="cuda"); t2 = torch.rand_like(t1)
    e1 = ops.hash_fwd_f32(x01, t1, lv); e2 = ops.hash_fwd_f32(x01, t2, lv); e12 = ops.hash_fwd_f32(x01, t1 + t2, lv)
    torch.testing.assert_close(e12, e1 + e2, rtol=1e-5, atol=1e-5)
    # <E(table), g> == <table, E^T(g)>
    g = torch.randn_like(e1)
    dt = torch.zeros_like(t1)
    ops.hash_bwd_f32(x01, g, lv, dt)
    lhs = (e1.double() * g.double()).sum().item(); rhs = (t1.double() * dt.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


def test_march_in_kernel_jitter(oracle, lego_batch):
    """Round 4: the trainer's march draws its per-ray jitter in the kernel (ngp_march_train_fused_rng: splitmix64 of (seed, ray) ->
    24-bit uniform) instead of reading a torch.rand tensor (ray_march.py:138 draws torch.rand_like -- any i.i.d. uniform in [0, 1)
    is the same algorithm).  (a) the values are uniform: moments, a Kolmogorov-Smirnov bound, no repeats across seeds / rays;
    (b) the march with seed s == the march given the explicit vector rng_uniform(s, .) == the oracle on that vector, bit for bit."""
    o, d, _, bits = lego_batch
    n = 4096
    u = ops.rng_uniform(1234567, 1 << 20).cpu().numpy().astype(np.float64)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1.0 / 12.0) < 1e-3
    ks = np.abs(np.sort(u) - (np.arange(u.size) + 0.5) / u.size).max()
    assert ks < 2.0 / np.sqrt(u.size)                                      # ~1.95 / sqrt(n) is the 0.1 % point
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5e-3                    # neighbouring rays are uncorrelated ...
    u2 = ops.rng_uniform(1234568, 1 << 20).cpu().numpy().astype(np.float64)
    assert abs(np.corrcoef(u, u2)[0, 1]) < 5e-3                            # ... and so are consecutive seeds
    seed = 987654321
    noise = ops.rng_uniform(seed, n)
    g_rng = ops.march_train_fused(dev(o[:n]), dev(d[:n]), None, dev(bits), None, 1, 0.5, 0.0, 128, 1024, seed=seed)
    g_vec = ops.march_train_fused(dev(o[:n]), dev(d[:n]), None, dev(bits), noise, 1, 0.5, 0.0, 128, 1024)
    hits = oracle.ray_aabb(o[:n], d[:n], 0.5)
    ref = oracle.march_train(o[:n], d[:n], hits, bits, noise.cpu().numpy(), 1, 0.5, 0.0, 128, 1024)
    _check_fused_march(ref, g_rng, n)
    _check_fused_march(ref, g_vec, n)


def test_march_fused_capacity_is_respected(oracle, lego_batch):
    """output arrays smaller than the data-dependent total must not be overrun -- what does not fit is dropped, the
    total still reports everything (the caller compares)."""
    o, d, noise, bits = lego_batch
    n = 2048
    hits = oracle.ray_aabb(o[:n], d[:n], 0.5)
    ref = oracle.march_train(o[:n], d[:n], hits, bits, noise[:n], 1, 0.5, 0.0, 128, 1024)
    total = ref[5]
    cap = total // 2
    guard = 4096
    from ngp_hip.ops import _lib, _ptr, _stream, MarchArena, coarse_bitfield, check
    import torch
    dv = torch.device("cuda")
    xyzs = torch.full((cap + guard, 3), -7.0, device=dv); dirs = torch.full((cap + guard, 3), -7.0, device=dv)
    deltas = torch.full((cap + guard,), -7.0, device=dv); ts = torch.full((cap + guard,), -7.0, device=dv)
    rays_a = torch.empty(n, 3, device=dv, dtype=torch.int32); tot = torch.zeros(1, device=dv, dtype=torch.int32)
    ctr = torch.zeros(2, device=dv, dtype=torch.int32)
    stage = MarchArena.get(dv, n, 1024)
    o_d, d_d, bits_d, noise_d = dev(o[:n]), dev(d[:n]), dev(bits), dev(noise[:n])      # (named: they must outlive the launch)
    coarse = coarse_bitfield(bits_d, 1, 128)
    check(_lib().ngp_march_train_fused_cap(_ptr(o_d), _ptr(d_d), _ptr(None), _ptr(bits_d), _ptr(coarse), _ptr(noise_d), 1, 128, 0.5, 0.0,
                                           1024, n, cap, _ptr(stage), _ptr(ctr), _ptr(rays_a), _ptr(tot), _ptr(xyzs), _ptr(dirs),
                                           _ptr(deltas), _ptr(ts), _stream()), "cap")
    assert int(tot[0]) == total > cap
    for t in (xyzs, dirs, deltas, ts):
        assert bool((t[cap:] == -7.0).all())                                # nothing beyond the capacity was touched
    assert bool((ts[:cap] != -7.0).all())                                   # everything below it was written
