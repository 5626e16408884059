"""The HIP kernels (through the C ABI) directly against the reference-executed golden vectors (tests/golden/ref_*.npz,
made by oracle/gen_golden.py from the reference's own kernel source)."""
import numpy as np
import pytest
import torch

from test_golden import G, beq, golden_table
from ngp_hip import ops, synthetic

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ray_aabb_golden(hip_lib):
    g = G("ref_ray_aabb.npz")
    for scale in (0.5, 16.0):
        assert beq(ops.ray_aabb(dev(g["rays_o"]), dev(g["rays_d"]), scale).cpu().numpy(), g["hits_%g" % scale])


@pytest.mark.parametrize("name,cascades,scale,esf,max_samples", [("ref_march_lego.npz", 1, 0.5, 0.0, 1024),
                                                                  ("ref_march_garden.npz", 6, 16.0, 1.0 / 256, 64)])
def test_march_train_golden(hip_lib, lego_bitfield, name, cascades, scale, esf, max_samples):
    g = G(name)
    bits = lego_bitfield if cascades == 1 else synthetic.ball_slab_bitfield(6, 16.0, seed=int(g["bitfield_seed"]))
    r = ops.march_train(dev(g["rays_o"]), dev(g["rays_d"]), dev(g["hits_t"]), dev(bits), dev(g["noise"]), cascades, scale, esf,
                        128, max_samples)
    assert int(r[5]) == int(g["total"]) and np.array_equal(r[0].cpu().numpy(), g["rays_a"])
    assert beq(r[4].cpu().numpy(), g["ts"]) and beq(r[3].cpu().numpy(), g["deltas"])
    assert beq(r[1].cpu().numpy(), g["xyzs"]) and beq(r[2].cpu().numpy(), g["dirs"])


def test_march_test_golden(hip_lib, lego_bitfield):
    g = G("ref_march_test.npz")
    hits = ops.ray_aabb(dev(g["rays_o"]), dev(g["rays_d"]), 0.5)
    for k in range(2):
        n_step = int(g["r%d_n_step" % k])
        r_idx, valid, deltas, ts, cnt = ops.march_test(dev(g["rays_o"]), dev(g["rays_d"]), hits, dev(g["alive"]), dev(lego_bitfield),
                                                       1, 0.5, 0.0, 128, n_step)
        m = g["r%d_valid" % k].astype(bool)
        assert np.array_equal(valid.cpu().numpy(), g["r%d_valid" % k]) and np.array_equal(cnt.cpu().numpy(), g["r%d_counter" % k])
        assert np.array_equal(r_idx.cpu().numpy()[m], g["r%d_ray_indices" % k][m])
        assert beq(ts.cpu().numpy()[m], g["r%d_ts" % k][m]) and beq(deltas.cpu().numpy()[m], g["r%d_deltas" % k][m])
        assert beq(hits.cpu().numpy(), g["r%d_hits" % k])


@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_hash_f32_golden(hip_lib, tag):
    g = G("ref_hash_f32_%s.npz" % tag)
    lv = ops.make_levels(2**19, 16, 16.0, float(g["max_res"]), 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])          # the scales the reference kernel evaluated (see test_golden.py)
    out = ops.hash_fwd_f32(dev(g["xyzs"]), dev(golden_table(int(g["total_param_size"]))), lv)
    assert beq(out.cpu().numpy(), g["out"])


@pytest.mark.parametrize("fixture", ["ref_hash_f16.npz", "ref_hash_f16_big.npz"])
def test_hash_f16_golden(hip_lib, oracle, fixture, monkeypatch):
    """Forward: bit-exact against the reference-executed vectors.  Backward (packed-f16-atomic kernel AND the LDS-sliced form):
    same touched rows; bit-exact on every row that receives a single contribution (its value does not depend on the order of
    the reference's atomics), within a few f16 ulp of the reference's serial-order result on shared rows."""
    g = G(fixture)
    lv = ops.make_levels(2**19, 16, 16.0, 1024.0, 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    table_h = golden_table(int(g["total_entries"]) * 2, -0.1, 0.1).astype(np.float16).reshape(-1, 2)
    out = ops.hash_fwd_f16(dev(g["xyzs"]), dev(table_h), lv).cpu().numpy()
    assert np.array_equal(out.view(np.uint16), g["out"].view(np.uint16))
    _, count = oracle.hash_bwd_f16_serial(g["xyzs"], g["dout"], lv)          # contributions per row (the checker, not the product)
    rows = g["grad_rows"]
    one = count[rows] == 1
    monkeypatch.setenv("NGP_HASH_BWD", "atomic")
    grad = torch.zeros(table_h.shape[0], 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(dev(g["xyzs"]), dev(g["dout"]), lv, grad)
    n = g["xyzs"].shape[0]
    sliced = torch.zeros(table_h.shape[0], 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16_sliced(dev(g["xyzs"]), dev(g["dout"]).float().reshape(n, -1).contiguous(), lv, sliced)
    # rows shared by several contributions depend on the order of the f16 adds (the reference's atomics, the HIP kernel's, the
    # fixture's serial order): every add rounds to f16, i.e. loses at most 2^-11 of the running sum, which never exceeds the sum of
    # the |contributions| -- the oracle's exact scatter-add of |dout| (the weights are non-negative).  Two such results differ by
    # at most twice count * 2^-11 * sum|contributions|.
    sum_abs = oracle.hash_bwd_f16(g["xyzs"], np.abs(g["dout"].astype(np.float32)).astype(np.float16), lv)[rows]
    bound = 2.0 * count[rows][:, None] * 2.0**-11 * sum_abs + 1e-7
    for name, t in (("atomic", grad), ("sliced", sliced)):
        got = t.cpu().numpy()
        assert np.array_equal(np.flatnonzero(np.abs(got.astype(np.float32)).sum(1)), rows), name
        assert np.array_equal(got[rows][one].view(np.uint16), g["grad_vals"][one].view(np.uint16)), name
        err = np.abs(got[rows].astype(np.float32) - g["grad_vals"].astype(np.float32))
        assert (err <= bound).all(), (name, float((err / bound).max()))


def test_sh16_and_grid_utils_golden(hip_lib):
    g = G("ref_sh16.npz")
    assert beq(ops.sh16_fwd(dev(g["dirs"])).cpu().numpy(), g["out"])
    g = G("ref_grid_utils.npz")
    assert np.array_equal(ops.morton3d(dev(g["coords"])).cpu().numpy(), g["morton"])
    assert np.array_equal(ops.morton3d_invert(dev(g["morton"])).cpu().numpy(), g["inverted"])
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ops.packbits(dev(g["grid"]), float(g["threshold"]), out)
    assert np.array_equal(out.cpu().numpy(), g["bitfield"])


def test_composites_golden(hip_lib):
    g = G("ref_composite_train.npz")
    tot, op, dep, rgb, ws = ops.composite_train_fwd(dev(g["sigmas"]), dev(g["rgbs"]), dev(g["deltas"]), dev(g["ts"]), dev(g["rays_a"]),
                                                    1e-4)
    assert np.abs(tot.cpu().numpy() - g["total_samples"]).max() <= 1
    np.testing.assert_allclose(op.cpu().numpy(), g["opacity"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dep.cpu().numpy(), g["depth"], rtol=1e-5, atol=1e-6)
    g = G("ref_composite_test.npz")
    alive, op, dep, rgb = dev(g["alive_in"]), dev(g["opacity_in"]), dev(g["depth_in"]), dev(g["rgb_in"])
    ops.composite_test(dev(g["sigmas"]), dev(g["rgbs"]), dev(g["deltas"]), dev(g["ts"]), dev(g["pack_info"]), alive, 1e-4, op, dep, rgb)
    assert np.array_equal(alive.cpu().numpy(), g["alive_out"])
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(op.cpu().numpy(), g["opacity_out"], rtol=1e-5, atol=1e-6)


def test_distortion_golden_and_oracle(hip_lib, oracle):
    g = G("ref_distortion.npz")
    loss, wi, wti = ops.distortion_fwd(dev(g["ws"]), dev(g["deltas"]), dev(g["ts"]), dev(g["rays_a"]))
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(wi.cpu().numpy(), g["ws_inc"], rtol=1e-5, atol=1e-8)
    dws = ops.distortion_bwd(dev(g["dL_dloss"]), dev(g["ws"]), dev(g["deltas"]), dev(g["ts"]), wi, wti, dev(g["rays_a"]))
    np.testing.assert_allclose(dws.cpu().numpy(), g["dL_dws"], rtol=1e-4, atol=1e-7)
    # larger seeded case against the oracle, through the autograd Function of modules/distortion.py
    from modules.distortion import distortion_loss
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 300, 2000).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays_a = np.stack([rng.permutation(2000).astype(np.int32), starts, counts], -1)
    S = int(counts.sum())
    ws = (rng.random(S, dtype=np.float32) * 0.05).astype(np.float32)
    deltas = np.full(S, 0.0017, np.float32)
    ts = (np.sort(rng.random(S, dtype=np.float32)) + 0.3).astype(np.float32)
    ref_loss, ref_wi, ref_wti = oracle.distortion_fwd(ws, deltas, ts, rays_a)
    w = dev(ws).requires_grad_(True)
    out = distortion_loss({"ws": w, "deltas": dev(deltas), "ts": dev(ts), "rays_a": dev(rays_a)})
    # the formula cancels two large prefix products: judge both f32 evaluations against float64
    f64 = np.zeros(2000)
    for r, s0, c in rays_a:
        wv, tv = ws[s0:s0 + c].astype(np.float64), ts[s0:s0 + c].astype(np.float64)
        wi, wti = np.cumsum(wv), np.cumsum(wv * tv)
        f64[r] = np.sum(2 * (wti * (wi - wv) - wi * (wti - wv * tv)) + wv * wv * deltas[s0:s0 + c] / 3)
    err_hip = np.abs(out.detach().cpu().numpy() - f64).max()
    err_ora = np.abs(ref_loss - f64).max()
    assert err_hip <= 3 * err_ora + 1e-6, (err_hip, err_ora)
    gl = rng.standard_normal(2000).astype(np.float32)
    out.backward(dev(gl))
    ref_dws = oracle.distortion_bwd(gl, deltas, ws, ts, ref_wi, ref_wti, rays_a)
    d64 = np.zeros(S)
    for r, s0, c in rays_a:
        if c == 0:
            continue
        wv, tv, dv = ws[s0:s0 + c].astype(np.float64), ts[s0:s0 + c].astype(np.float64), deltas[s0:s0 + c].astype(np.float64)
        wi, wti = np.cumsum(wv), np.cumsum(wv * tv)
        sel = np.concatenate([[0.0], tv[1:] * wi[:-1] - wti[:-1]])
        d64[s0:s0 + c] = gl[r] * 2 * (sel + (wti[-1] - wti - tv * (wi[-1] - wi))) + gl[r] * (2.0 / 3.0) * wv * dv
    err_hip = np.abs(w.grad.cpu().numpy() - d64).max()
    err_ora = np.abs(ref_dws - d64).max()
    assert err_hip <= 3 * err_ora + 1e-6 * np.abs(d64).max(), (err_hip, err_ora)


# ---------------------------------------------------------------------------------------------- f-3 against the reference's torch code
class _Grid:
    """The attributes OccupancyUpdater / NGP.mark_invisible_cells read, for a small grid (the fixtures use 2 cascades of 16^3)."""

    def __init__(self, g, grid):
        from modules.networks import NGP, _cell_coords
        import types
        self.grid_size, self.cascades, self.scale = int(g["grid_size"]), int(g["cascades"]), float(g["scale"])
        self.density_grid = dev(grid).clone()
        self.density_bitfield = torch.zeros(self.cascades * self.grid_size**3 // 8, dtype=torch.uint8, device="cuda")
        self.grid_coords = _cell_coords(self.grid_size).cuda()
        self.get_all_cells = types.MethodType(NGP.get_all_cells, self)


def test_update_density_grid_vs_reference(hip_lib):
    """ngp_hip/occupancy.py held to vectors produced by the reference's OWN NGP.update_density_grid /
    sample_uniform_and_occupied_cells / get_all_cells (modules/networks.py:168-209,255-290; oracle/gen_golden_r3.py) on a
    2-cascade 16^3 grid: the recorded random draws go in through the updater's hooks, the density is the reference-evaluated
    analytic field (the density network is not what is under test).  Cell positions, the decay/max merge, the invisible cells
    and the packed bitfield must come out bit for bit."""
    from ngp_hip.occupancy import OccupancyUpdater
    g = G("ref_update_density_grid.npz")
    G3 = int(g["grid_size"])**3
    thr, decay = float(g["threshold"]), float(g["decay"])
    # ---- warm-up: all cells; the recorded jitter is in grid_coords order, the updater enumerates cells by Morton code
    m = _Grid(g, g["grid_before"])
    upd = OccupancyUpdater(m)
    order = g["all_indices"].astype(np.int64)
    seen = []

    def jitter(c, n):
        j = np.empty((G3, 3), np.float32)
        j[order] = g["warm_jitter"][c]
        return dev(j)

    def density(c, xyzs, indices):
        assert indices is None
        seen.append(beq(xyzs.cpu().numpy(), g["warm_xyz_by_morton"][c]))
        return dev(g["warm_sigma_by_morton"][c])
    upd.update(thr, warmup=True, decay=decay, jitter=jitter, density_fn=density)
    assert seen == [True] * m.cascades                                           # jittered cell positions, bit-exact
    assert beq(m.density_grid.cpu().numpy(), g["warm_grid_after"])
    assert np.array_equal(m.density_bitfield.cpu().numpy(), g["warm_bitfield"])
    # ---- sampled update: the reference's randint draws, expressed as the uniforms that select the same cells
    m = _Grid(g, g["samp_grid_before"])
    upd = OccupancyUpdater(m)

    def uniforms(c):
        code = ops.morton3d(dev(g["samp_coords1"][c].astype(np.int32))).double()
        u_cell = ((code + 0.5) / G3).float()
        u_pick = ((dev(g["samp_rand_idx"][c]).double() + 0.5) / max(int(g["samp_n_occupied"][c]), 1)).float()
        return u_cell, u_pick

    def density_c(c, xyzs, indices):
        return dev(g["centre_sigma_by_morton"][c])[indices.long()]
    upd.update(thr, warmup=False, decay=decay, jitter=lambda c, n: torch.full((n, 3), 0.5, device="cuda"), uniforms=uniforms,
               density_fn=density_c)
    assert beq(m.density_grid.cpu().numpy(), g["samp_grid_after"])
    assert np.array_equal(m.density_bitfield.cpu().numpy(), g["samp_bitfield"])


def test_mark_invisible_cells_vs_reference(hip_lib):
    """modules.networks.NGP.mark_invisible_cells against the reference's (modules/networks.py:212-253) on the fixture grid: a
    cell's flag may only differ where a projection lands within float rounding of an image border or the near plane."""
    from modules.networks import NGP
    g = G("ref_mark_invisible.npz")
    m = _Grid(g, np.zeros_like(g["density_grid"]))
    NGP.mark_invisible_cells(m, dev(g["K"]), dev(g["poses"]), tuple(int(v) for v in g["img_wh"]), chunk=1000)
    got, want = m.density_grid.cpu().numpy(), g["density_grid"]
    assert set(np.unique(got)) <= {-1.0, 0.0} and 0.2 < (want < 0).mean() < 0.5
    assert (got != want).mean() < 2e-3, (got != want).mean()
    np.testing.assert_allclose(m.count_grid.cpu().numpy(), g["count_grid"], atol=1.0 / g["poses"].shape[0] + 1e-6)
    assert (np.abs(m.count_grid.cpu().numpy() - g["count_grid"]) > 1e-6).mean() < 2e-3


# ---- round 4: the HIP backward kernels against the vectors derived from the reference's source (oracle/gen_golden_autodiff.py:
# the reference's autograd glue + forward kernels under the shim's reverse-mode tape) -- a-4, a-6, a-7 backward
@pytest.mark.parametrize("tag,max_res", [("c2", 1024.0), ("c3", 4096.0)])
@pytest.mark.parametrize("form", ["atomic", "sliced"])
def test_hash_bwd_f32_autodiff_golden(hip_lib, tag, max_res, form):
    g = G("ref_hash_f32_%s_grad.npz" % tag)
    lv = ops.make_levels(2**19, 16, 16.0, max_res, 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    n_ent = int(g["total_entries"])
    dtable = torch.zeros(n_ent * 2, device="cuda")
    x, dout = dev(g["xyzs"]), dev(g["dout"])
    if form == "sliced":
        ops.hash_bwd_f32_sliced(x, dout, lv, dtable)
    else:
        import ctypes
        ops.check(ops._lib().ngp_hash_bwd_f32(ops._ptr(x), ops._ptr(dout), ctypes.byref(lv), x.shape[0], ops._ptr(dtable), ops._stream()),
                  "ngp_hash_bwd_f32")
    grad = dtable.cpu().numpy().reshape(-1, 2)
    rows = np.flatnonzero((grad != 0).any(1))
    assert np.array_equal(rows, g["grad_rows"])
    ref = g["grad_vals"]
    assert np.abs(grad[rows] - ref).max() <= 1e-6 * np.abs(ref).max()
    np.testing.assert_allclose(grad[rows], ref, rtol=2e-5, atol=1e-6 * np.abs(ref).max())


def test_sh16_bwd_autodiff_golden(hip_lib):
    g = G("ref_sh16_grad.npz")
    dd = ops.sh16_bwd(dev(g["dirs"]), dev(g["dout"])).cpu().numpy()
    np.testing.assert_allclose(dd, g["ddirs"], rtol=0, atol=1e-6 * np.abs(g["ddirs"]).max())


def test_composite_train_bwd_autodiff_golden(hip_lib):
    g = G("ref_composite_train_grad.npz")
    S = int(g["n_valid"])
    sig, rgbs, dl, ts, ra = dev(g["sigmas"]), dev(g["rgbs"]), dev(g["deltas"]), dev(g["ts"]), dev(g["rays_a"])
    tot, op, dep, rgb, ws = ops.composite_train_fwd(sig, rgbs, dl, ts, ra, 1e-4)
    np.testing.assert_allclose(op.cpu().numpy(), g["opacity"], rtol=1e-5, atol=1e-6)
    ds, dc = ops.composite_train_bwd(dev(g["g_opacity"]), dev(g["g_depth"]), dev(g["g_rgb"]), dev(g["g_ws"]), sig, rgbs, dl, ts, ra,
                                     op, dep, rgb, ws, 1e-4)
    ds, dc = ds.float().cpu().numpy(), dc.float().cpu().numpy()
    # wave-scan order of the suffix sums vs the tape's serial order: 1e-5 of the largest entry
    np.testing.assert_allclose(ds[:S], g["d_sigmas"][:S], rtol=0, atol=1e-5 * np.abs(g["d_sigmas"]).max())
    np.testing.assert_allclose(dc[:S], g["d_rgbs"][:S], rtol=0, atol=1e-5 * np.abs(g["d_rgbs"]).max())
