"""Known-answer tests of the CPU oracle: the reference ships no tests, so the oracle
carries its own hand-computed cases; gradients are checked against torch autograd of a plain restatement."""
import numpy as np
import pytest
import torch

from ngp_hip import synthetic


def test_frexp_bit_kat(oracle):
    # bit-level emulation of reference modules/utils.py:60-75; differs from C frexp exactly on powers of two
    for x, e in [(0.0, 0), (0.125, -3), (0.3, -1), (0.5, -1), (1.0, 0), (1.5, 1), (2.0, 1), (127.99, 7),
                 (0.21650635, -2), (2.0, 1), (8.0, 3)]:
        assert oracle.frexp_bit(x) == e, x


def test_morton_kat_and_roundtrip(oracle):
    pts = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 6], [127, 127, 127]], np.int32)
    assert oracle.morton3d(pts).tolist() == [1, 2, 4, 427, 2097151]
    g = np.arange(128, dtype=np.int32)
    allc = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    idx = oracle.morton3d(allc)
    assert np.array_equal(np.sort(idx), np.arange(128**3))          # a bijection onto [0, 128^3)
    assert np.array_equal(oracle.morton3d_invert(idx), allc)


def test_packbits_kat(oracle):
    grid = np.array([0.0, 1.0, 0.5, 2.0, -1.0, 0.51, 0.5, 9.0] * 2, np.float32)
    assert oracle.packbits(grid, 0.5).tolist() == [0b10101010, 0b10101010]


def test_level_table_matches_reference_printout(oracle):
    # notebooks/pipeline.ipynb cell 1 output: offset_ 5710032, total_hash_size 11420064 (Lego, max_res 1024)
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    assert list(lv.resolution[:16]) == [16, 22, 28, 37, 49, 64, 85, 112, 148, 195, 256, 338, 446, 589, 777, 1024]
    assert list(lv.map_size[:6]) == [4096, 10648, 21952, 50656, 117656, 262144] and set(lv.map_size[6:16]) == {524288}
    assert lv.total_entries == 5710032 and lv.begin_fast_hash_level == 6
    lv = oracle.make_levels(2**19, 16, 16, 4096, 2)
    assert list(lv.resolution[:16]) == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]
    assert list(lv.map_size[:5]) == [4096, 13824, 39304, 117656, 357912]
    assert lv.total_entries == 6299960 and lv.begin_fast_hash_level == 5


def test_hip_and_oracle_level_tables_agree(oracle, hip_lib):
    from ngp_hip import ops
    for cfg in [(2**19, 16, 16, 1024, 2), (2**19, 16, 16, 4096, 2), (2**21, 4, 32, 128, 4), (2**19, 16, 16, 2048, 2)]:
        assert bytes(oracle.make_levels(*cfg)) == bytes(ops.make_levels(*cfg))


def test_hash_lattice_corner_returns_feature(oracle):
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(0)
    table = rng.random(lv.total_entries * 2, dtype=np.float32)
    for level in (0, 6, 15):                      # dense, first hashed, finest
        scale = np.float32(lv.scale[level])
        cell = np.array([3, 5, 7], np.uint32)
        x = ((cell.astype(np.float32) - np.float32(0.5)) / scale).astype(np.float32)   # pos = x*scale+0.5 == cell
        pos = x * scale + np.float32(0.5)
        if not np.array_equal(np.floor(pos), cell.astype(np.float32)) or np.any(pos - np.floor(pos) != 0):
            continue                               # rounding moved it off the lattice; skip this level
        idx, w = oracle.hash_corners(x[None], lv)
        assert w[0, level, 0] == 1.0 and np.all(w[0, level, 1:] == 0.0)
        res = lv.resolution[level]
        if level < lv.begin_fast_hash_level:
            want = (3 + 5 * res + 7 * res * res) % lv.map_size[level]
        else:
            want = (3 ^ (5 * 2654435761 & 0xffffffff) ^ (7 * 805459861 & 0xffffffff)) % lv.map_size[level]
        assert idx[0, level, 0] == lv.offset[level] + want
        out = oracle.hash_fwd_f32(x[None], table, lv)
        assert np.array_equal(out[0, 2 * level:2 * level + 2], table[2 * idx[0, level, 0]:2 * idx[0, level, 0] + 2])


def test_hash_dense_wraparound_kat(oracle):
    """x = 1 on a dense level: cell+1 == res carries into the next row / past the level and wraps with % size."""
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    idx, w = oracle.hash_corners(np.array([[1.0, 1.0, 1.0]], np.float32), lv)
    res, size = lv.resolution[0], lv.map_size[0]
    cell = int(np.floor(np.float32(1.0) * np.float32(lv.scale[0]) + np.float32(0.5)))
    want = [((cell + (c & 1)) + (cell + ((c >> 1) & 1)) * res + (cell + ((c >> 2) & 1)) * res * res) % size for c in range(8)]
    assert idx[0, 0].tolist() == want
    assert np.isclose(w[0, 0].sum(), 1.0, atol=1e-6)


def _torch_hash(x, table, lv):
    """vectorised fp64 torch restatement of the trilinear gather, for autograd ground truth."""
    from oracle import ngp_oracle as ora
    idx, w = ora.hash_corners(x, lv)
    idx_t = torch.from_numpy(idx.astype(np.int64))
    w_t = torch.from_numpy(w.astype(np.float64))
    tab = table.view(-1, lv.n_features)
    return (tab[idx_t] * w_t[..., None]).sum(2).reshape(x.shape[0], -1)


def test_hash_fwd_bwd_vs_torch_autograd(oracle):
    lv = oracle.make_levels(2**14, 8, 16, 256, 2)
    rng = np.random.default_rng(1)
    x = rng.random((4096, 3), dtype=np.float32)
    table = torch.rand(lv.total_entries * 2, dtype=torch.float64, requires_grad=True)
    ref = _torch_hash(x, table, lv)
    out = oracle.hash_fwd_f32(x, table.detach().float().numpy(), lv)
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    g = rng.standard_normal(out.shape).astype(np.float32)
    ref.backward(torch.from_numpy(g.astype(np.float64)))
    np.testing.assert_allclose(oracle.hash_bwd_f32(x, g, lv), table.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_sh16_bwd_vs_autograd(oracle):
    rng = np.random.default_rng(2)
    d = torch.from_numpy(rng.random((256, 3))).requires_grad_(True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    e = torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.94617469575755997 * z * z - 0.31539156525251999,
        -1.0925484305920792 * x * z, 0.54627421529603959 * x * x - 0.54627421529603959 * y * y,
        0.59004358992664352 * y * (-3.0 * x * x + y * y), 2.8906114426405538 * x * y * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z * z), 0.3731763325901154 * z * (5.0 * z * z - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z * z), 1.4453057213202769 * z * (x * x - y * y),
        0.59004358992664352 * x * (-x * x + 3.0 * y * y)], 1)
    np.testing.assert_allclose(oracle.sh16_fwd(d.detach().numpy()), e.detach().numpy(), rtol=1e-5, atol=1e-6)
    g = torch.from_numpy(rng.standard_normal((256, 16)))
    e.backward(g)
    np.testing.assert_allclose(oracle.sh16_bwd(d.detach().numpy(), g.numpy()), d.grad.numpy(), rtol=1e-5, atol=1e-5)


def _torch_composite(sig, rgbs, deltas, ts, rays_a, thr):
    n = rays_a.shape[0]
    op = [None] * n; dep = [None] * n; rgb = [None] * n
    ws = torch.zeros_like(sig)
    ws_list = []
    for row in range(n):
        r, start, N = [int(v) for v in rays_a[row]]
        T = torch.ones((), dtype=sig.dtype)
        o = torch.zeros((), dtype=sig.dtype); dd = torch.zeros((), dtype=sig.dtype); c = torch.zeros(3, dtype=sig.dtype)
        for j in range(N):
            s = start + j
            if T.item() > thr:
                a = 1.0 - torch.exp(-sig[s] * deltas[s])
                w = a * T
                c = c + w * rgbs[s]; dd = dd + w * ts[s]; o = o + w
                ws_list.append((s, w))
                T = T * (1.0 - a)
        op[r], dep[r], rgb[r] = o, dd, c
    return torch.stack(op), torch.stack(dep), torch.stack(rgb), ws_list


@pytest.mark.parametrize("with_ws_grad", [False, True])
def test_composite_closed_form_bwd_vs_autograd(oracle, with_ws_grad):
    rng = np.random.default_rng(3)
    counts = np.array([0, 5, 40, 1, 17, 0, 64, 9], np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays_a = np.stack([np.array([3, 0, 7, 1, 6, 2, 4, 5], np.int32), starts, counts], -1)
    S = int(counts.sum())
    sig_np = (rng.random(S) * 40).astype(np.float32); sig_np[starts[2]:starts[2] + 40] *= 20   # forces early termination
    rgbs_np = rng.random((S, 3)).astype(np.float32)
    deltas = np.full(S, 0.0169, np.float32); ts_np = (rng.random(S) + 0.5).astype(np.float32)
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(sig_np, rgbs_np, deltas, ts_np, rays_a, 1e-4)
    assert tot[7] < 40                                    # ray_idx 7 (row 2) terminated early
    sig = torch.from_numpy(sig_np.astype(np.float64)).requires_grad_(True)
    rgbs = torch.from_numpy(rgbs_np.astype(np.float64)).requires_grad_(True)
    t_op, t_dep, t_rgb, ws_list = _torch_composite(sig, rgbs, torch.from_numpy(deltas.astype(np.float64)),
                                                   torch.from_numpy(ts_np.astype(np.float64)), rays_a, 1e-4)
    np.testing.assert_allclose(op, t_op.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb, t_rgb.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dep, t_dep.detach().numpy(), rtol=1e-5, atol=1e-6)
    g_op = rng.standard_normal(8); g_dep = rng.standard_normal(8); g_rgb = rng.standard_normal((8, 3))
    g_ws = rng.standard_normal(S) if with_ws_grad else None
    loss = (t_op * torch.from_numpy(g_op)).sum() + (t_dep * torch.from_numpy(g_dep)).sum() + (t_rgb * torch.from_numpy(g_rgb)).sum()
    if with_ws_grad:
        for s, w in ws_list:
            loss = loss + g_ws[s] * w
    loss.backward()
    ds, dc = oracle.composite_train_bwd(g_op, g_dep, g_rgb, g_ws, sig_np, rgbs_np, deltas, ts_np, rays_a, 1e-4)
    np.testing.assert_allclose(ds, sig.grad.numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(dc, rgbs.grad.numpy(), rtol=2e-4, atol=1e-7)


def test_single_ray_two_cell_grid(oracle):
    """One +x ray through an otherwise empty grid with two occupied cells: exact count and orbit."""
    G = 128
    bits = np.zeros(G**3 // 8, np.uint8)
    cells = [(70, 64, 64), (71, 64, 64)]
    for c in cells:
        m = int(oracle.morton3d(np.array([c], np.int32))[0])
        bits[m >> 3] |= 1 << (m & 7)
    o = np.array([[-1.0, 0.001, 0.001]], np.float32); d = np.array([[1.0, 0.0, 0.0]], np.float32)
    hits = oracle.ray_aabb(o, d, 0.5)
    assert hits[0, 0] == 0.5 and hits[0, 1] == 1.5
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, np.zeros(1, np.float32), 1, 0.5, 0.0, G, 1024)
    dt = np.float32(1.7320508075688772 / 1024)
    # independent float32 emulation of ray_march.py:46-74 for this axis-aligned ray (y, z never limit the skip:
    # their direction component is 0 and the cell centre lies at +inf)
    f = np.float32
    t = f(0.5); want = []
    while t < f(1.5):
        x = f(-1.0) + t * f(1.0)
        n = min(max(f(0.5) * (x * f(2.0) + f(1.0)) * f(G), f(0)), f(G - 1))
        if int(n) in (70, 71):
            want.append(t)
            t = f(t + dt)
        else:
            tx = (((f(f(n + f(0.5)) + f(0.5)) * f(1.0 / G)) * f(2.0) - f(1.0)) * f(0.5) - x) * f(1.0)
            t_target = f(t + max(f(0), tx))
            t = f(t + dt)
            while t < t_target:
                t = f(t + dt)
    assert total == len(want) and 5 <= total <= 10          # 2 cells / (dt*128) = 9.24 steps, minus skip overshoot
    assert np.array_equal(ts, np.array(want, np.float32))
    assert np.all(deltas == dt) and np.array_equal(rays_a, [[0, 0, total]])
    assert np.array_equal(dirs, np.repeat(d, total, 0))


def test_march_lego_statistics(oracle, lego_bitfield):
    o, d = synthetic.lego_rays(4096, seed=23)
    hits = oracle.ray_aabb(o, d, 0.5)
    noise = np.random.default_rng(0).random(4096, dtype=np.float32)
    rays_a, total = oracle.march_train(o, d, hits, lego_bitfield, noise, 1, 0.5, 0.0, 128, 1024, count_only=True)
    assert np.array_equal(rays_a[:, 0], np.arange(4096))
    assert np.array_equal(rays_a[:, 1], np.concatenate([[0], np.cumsum(rays_a[:, 2])[:-1]]))
    assert 10 < total / 4096 < 45                            # SURVEY probe: ~19.8 (uniform pixels) .. 40.7 (centred)


def test_march_test_progressive_equals_train_march(oracle, lego_bitfield):
    """Appendix B.9 (march half): resumed raymarching_test rounds visit exactly the train march's samples (no jitter)."""
    o, d = synthetic.lego_rays(512, seed=5)
    hits = oracle.ray_aabb(o, d, 0.5)
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, lego_bitfield, np.zeros(512, np.float32), 1, 0.5, 0.0,
                                                              128, 1024)
    h = hits.copy()
    alive = np.arange(512, dtype=np.int64)
    got = [[] for _ in range(512)]
    for _ in range(40):
        r_idx, valid, dl, t, cnt = oracle.march_test(o, d, h, alive, lego_bitfield, 1, 0.5, 0.0, 128, 32)
        m = valid.astype(bool)
        for r, tt in zip(r_idx[m], t[m]):
            got[r].append(tt)
        alive = alive[cnt == 32]
        if len(alive) == 0:
            break
    for r in range(512):
        s, n = rays_a[r, 1], rays_a[r, 2]
        # hits_t[:,0] >= 0.01 > 0 so the strict `0 < t` start condition of the test kernel does not bite
        assert np.array_equal(np.array(got[r], np.float32), ts[s:s + n])
