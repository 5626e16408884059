"""Pinning: the CPU oracle (oracle/ngp_oracle.c) against golden vectors produced by executing the reference's own
kernel source under oracle/ti_shim (oracle/gen_golden.py; committed as tests/golden/ref_*.npz).

Bit-exact wherever the arithmetic is +,-,*,/ and integer ops (ray-AABB, the whole march, SH, Morton, packbits, the
distortion scans); tolerance (1e-6 .. 1e-5) only where a transcendental is involved (expf in the compositor; the
per-level exp() of the hash grid, which moves `scale` by at most 1 ulp between libm implementations -- SURVEY H2)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ngp_hip import synthetic


def G(name):
    return np.load(os.path.join(GOLDEN, name))


def beq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def golden_table(n, lo=0.0, hi=1.0):          # must mirror oracle/gen_golden.py
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(2**32)
    return (lo + (hi - lo) * (h.astype(np.float64) / 2**32)).astype(np.float32)


def test_ray_aabb_vs_reference(oracle):
    g = G("ref_ray_aabb.npz")
    for scale in (0.5, 16.0):
        assert beq(oracle.ray_aabb(g["rays_o"], g["rays_d"], scale), g["hits_%g" % scale])


@pytest.mark.parametrize("name,cascades,scale,esf,max_samples", [("ref_march_lego.npz", 1, 0.5, 0.0, 1024),
                                                                  ("ref_march_garden.npz", 6, 16.0, 1.0 / 256, 64)])
def test_march_train_vs_reference(oracle, lego_bitfield, name, cascades, scale, esf, max_samples):
    g = G(name)
    bits = lego_bitfield if cascades == 1 else synthetic.ball_slab_bitfield(6, 16.0, seed=int(g["bitfield_seed"]))
    assert beq(oracle.ray_aabb(g["rays_o"], g["rays_d"], scale), g["hits_t"])
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(g["rays_o"], g["rays_d"], g["hits_t"], bits, g["noise"], cascades,
                                                              scale, esf, 128, max_samples)
    assert total == int(g["total"]) and total > 500
    assert np.array_equal(rays_a, g["rays_a"])          # serial Taichi execution == ray order == our prefix sum
    assert beq(ts, g["ts"]) and beq(deltas, g["deltas"]) and beq(xyzs, g["xyzs"]) and beq(dirs, g["dirs"])
    if max_samples == 64:
        assert rays_a[:, 2].max() == 64                  # the truncation branch is exercised


def test_march_test_vs_reference(oracle, lego_bitfield):
    g = G("ref_march_test.npz")
    hits = oracle.ray_aabb(g["rays_o"], g["rays_d"], 0.5)
    for k in range(2):
        n_step = int(g["r%d_n_step" % k])
        r_idx, valid, deltas, ts, cnt = oracle.march_test(g["rays_o"], g["rays_d"], hits, g["alive"], lego_bitfield, 1, 0.5, 0.0,
                                                          128, n_step)
        m = valid.astype(bool)
        assert np.array_equal(valid, g["r%d_valid" % k]) and np.array_equal(cnt, g["r%d_counter" % k])
        assert np.array_equal(r_idx[m], g["r%d_ray_indices" % k][m])
        assert beq(ts[m], g["r%d_ts" % k][m]) and beq(deltas[m], g["r%d_deltas" % k][m])
        assert beq(hits, g["r%d_hits" % k])              # in-place resume state
    assert m.sum() > 10


@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_hash_f32_vs_reference(oracle, tag):
    g = G("ref_hash_f32_%s.npz" % tag)
    lv = oracle.make_levels(2**19, 16, 16.0, float(g["max_res"]), 2)
    assert np.array_equal(np.array(lv.offset[:16]), g["offsets"].astype(np.uint32))
    assert np.array_equal(np.array(lv.map_size[:16]), g["hash_map_sizes"].astype(np.uint32))
    assert lv.begin_fast_hash_level == int(g["begin_fast_hash_level"])
    assert lv.total_entries * 2 == int(g["total_param_size"])
    table = golden_table(int(g["total_param_size"]))
    # with the oracle's own (glibc expf) level scales: equal up to the 1-ulp-of-scale effect on the finest levels
    np.testing.assert_allclose(oracle.hash_fwd_f32(g["xyzs"], table, lv), g["out"], rtol=1e-3, atol=1e-3)
    # with the scales the reference kernel actually evaluated: the gather itself is bit-exact
    assert np.abs(np.array(lv.scale[:16]) / g["scale_used"] - 1).max() < 2e-7
    assert np.array_equal(np.ceil(g["scale_used"]).astype(np.uint32) + 1, np.array(lv.resolution[:16]))
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    assert beq(oracle.hash_fwd_f32(g["xyzs"], table, lv), g["out"])


@pytest.mark.parametrize("fixture", ["ref_hash_f16.npz", "ref_hash_f16_big.npz"])
def test_hash_f16_vs_reference(oracle, fixture):
    """a-5, the half2 encoder against the reference's own kernels executed under the shim (Taichi-f16-exact since round 3: every
    f16 operation of the kernel is one correctly rounded operation, and `+=` casts its right-hand side to the table's f16 first,
    as Taichi's atomic add does).  Forward: bit-exact.  Backward: the oracle's serial-order form is bit-exact on EVERY row; its
    order-free form (exact sum, one rounding -- what the HIP kernels are held to) is bit-exact wherever a row receives a single
    contribution and within f16 accuracy of the serial result elsewhere."""
    g = G(fixture)
    lv = oracle.make_levels(2**19, 16, 16.0, 1024.0, 2)
    n_ent = int(g["total_entries"])
    table_h = golden_table(n_ent * 2, -0.1, 0.1).astype(np.float16).reshape(-1, 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    out = oracle.hash_fwd_f16(g["xyzs"], table_h, lv)
    assert out.dtype == np.float16 and np.array_equal(out.view(np.uint16), g["out"].view(np.uint16))
    serial, count = oracle.hash_bwd_f16_serial(g["xyzs"], g["dout"], lv)
    rows = np.flatnonzero(np.abs(serial.astype(np.float32)).sum(1))
    assert np.array_equal(rows, g["grad_rows"])
    assert np.array_equal(serial[rows].view(np.uint16), g["grad_vals"].view(np.uint16))
    grad = oracle.hash_bwd_f16(g["xyzs"], g["dout"], lv)                     # f64 accumulation, one rounding
    assert np.array_equal(np.flatnonzero(np.abs(grad).sum(1)), rows)
    one = count[rows] == 1
    assert one.mean() > 0.8
    assert np.array_equal(grad[rows][one].astype(np.float16).view(np.uint16), g["grad_vals"][one].view(np.uint16))
    # shared rows: the serial f16 accumulation loses at most 2^-11 of the running sum per add, and the running sum is bounded by the
    # sum of the |contributions| (= the exact scatter-add of |dout|: the weights are non-negative)
    sum_abs = oracle.hash_bwd_f16(g["xyzs"], np.abs(g["dout"].astype(np.float32)).astype(np.float16), lv)[rows]
    bound = count[rows][:, None] * 2.0**-11 * sum_abs + 2.0**-11 * np.abs(grad[rows]) + 1e-7
    assert (np.abs(grad[rows] - g["grad_vals"].astype(np.float32)) <= bound).all()


def test_sh16_vs_reference(oracle):
    g = G("ref_sh16.npz")
    assert beq(oracle.sh16_fwd(g["dirs"]), g["out"])


def test_composite_train_vs_reference(oracle):
    g = G("ref_composite_train.npz")
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(g["sigmas"], g["rgbs"], g["deltas"], g["ts"], g["rays_a"], 1e-4)
    assert np.array_equal(tot, g["total_samples"]) and tot.max() < 120      # early termination happened
    np.testing.assert_allclose(op, g["opacity"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(dep, g["depth"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(rgb, g["rgb"], rtol=2e-6, atol=1e-7)
    live = ~np.isnan(g["ws"])                   # the reference never writes ws behind early termination (torch.empty)
    np.testing.assert_allclose(ws[live], g["ws"][live], rtol=2e-6, atol=2e-7)   # 1 - exp(-x) cancels: absolute error
    assert np.all(ws[~live] == 0.0)             # ... the oracle / HIP kernels define those as 0


def test_composite_test_vs_reference(oracle):
    g = G("ref_composite_test.npz")
    alive, op, dep, rgb = g["alive_in"].copy(), g["opacity_in"].copy(), g["depth_in"].copy(), g["rgb_in"].copy()
    oracle.composite_test(g["sigmas"], g["rgbs"], g["deltas"], g["ts"], g["pack_info"], alive, 1e-4, op, dep, rgb)
    assert np.array_equal(alive, g["alive_out"]) and (alive < 0).any() and (alive >= 0).any()
    np.testing.assert_allclose(op, g["opacity_out"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(dep, g["depth_out"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(rgb, g["rgb_out"], rtol=2e-6, atol=1e-7)


def test_grid_utils_vs_reference(oracle):
    g = G("ref_grid_utils.npz")
    assert np.array_equal(oracle.morton3d(g["coords"]), g["morton"])
    assert np.array_equal(oracle.morton3d_invert(g["morton"]), g["inverted"]) and np.array_equal(g["inverted"], g["coords"])
    assert np.array_equal(oracle.packbits(g["grid"], float(g["threshold"])), g["bitfield"])


def test_distortion_vs_reference(oracle):
    g = G("ref_distortion.npz")
    loss, wi, wti = oracle.distortion_fwd(g["ws"], g["deltas"], g["ts"], g["rays_a"])
    assert beq(wi, g["ws_inc"]) and beq(wti, g["wts_inc"])
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-6, atol=1e-9)
    dws = oracle.distortion_bwd(g["dL_dloss"], g["deltas"], g["ws"], g["ts"], wi, wti, g["rays_a"])
    np.testing.assert_allclose(dws, g["dL_dws"], rtol=1e-6, atol=1e-9)


# ---- round 4: the three autodiff backwards against vectors DERIVED FROM THE REFERENCE'S SOURCE (oracle/gen_golden_autodiff.py: the
# reference's own autograd glue + forward kernels, `kernel.grad` emulated by a reverse-mode float32 tape under ti_shim).  The
# oracle's closed forms were derived by hand; this is what pins them to the reference.
@pytest.mark.parametrize("tag,max_res", [("c2", 1024.0), ("c3", 4096.0)])
def test_hash_bwd_f32_vs_reference_autodiff(oracle, tag, max_res):
    g = G("ref_hash_f32_%s_grad.npz" % tag)
    lv = oracle.make_levels(2**19, 16, 16.0, max_res, 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    table = golden_table(int(g["total_entries"]) * 2)
    assert beq(oracle.hash_fwd_f32(g["xyzs"], table, lv), g["out"])          # same forward as the taped run
    grad = oracle.hash_bwd_f32(g["xyzs"], g["dout"], lv).reshape(-1, 2)
    rows = np.flatnonzero((grad != 0).any(1))
    assert np.array_equal(rows, g["grad_rows"])                               # same touched entries
    ref = g["grad_vals"]
    # every contribution is ONE f32 product w * g on both sides; rows with several contributions differ by the order of the f32
    # adds only (the tape: reverse of the serial forward order; the oracle: f64 accumulation, one rounding)
    assert np.abs(grad[rows] - ref).max() <= 1e-6 * np.abs(ref).max()
    np.testing.assert_allclose(grad[rows], ref, rtol=2e-5, atol=1e-6 * np.abs(ref).max())
    # the reference's torch glue doubles the gradient of a LEAF parameter (hash_encoder.py:277): recorded, not copied
    assert float(g["module_leaf_factor"]) == 2.0


def test_sh16_bwd_vs_reference_autodiff(oracle):
    g = G("ref_sh16_grad.npz")
    dd = oracle.sh16_bwd(g["dirs"], g["dout"])
    np.testing.assert_allclose(dd, g["ddirs"], rtol=0, atol=1e-6 * np.abs(g["ddirs"]).max())


def test_composite_train_bwd_vs_reference_autodiff(oracle):
    g = G("ref_composite_train_grad.npz")
    S = int(g["n_valid"])                        # (one padding sample behind the last ray: the reference writes T[s + 1])
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(g["sigmas"], g["rgbs"], g["deltas"], g["ts"], g["rays_a"], 1e-4)
    np.testing.assert_allclose(op, g["opacity"], rtol=0, atol=5e-7); np.testing.assert_allclose(rgb, g["rgb"], rtol=0, atol=5e-7)   # (expf: libm ulp)
    ds, dc = oracle.composite_train_bwd(g["g_opacity"], g["g_depth"], g["g_rgb"], g["g_ws"], g["sigmas"], g["rgbs"], g["deltas"],
                                        g["ts"], g["rays_a"], 1e-4)
    # the gradient w.r.t. sigma is a suffix sum over the ray (up to 120 samples) evaluated in a different order: 1e-6 of the
    # largest entry absolute, like the forward's expf-bound tolerance
    np.testing.assert_allclose(ds[:S], g["d_sigmas"][:S], rtol=0, atol=2e-6 * np.abs(g["d_sigmas"]).max())
    np.testing.assert_allclose(dc[:S], g["d_rgbs"][:S], rtol=0, atol=1e-6 * np.abs(g["d_rgbs"]).max())
    assert not ds[S:].any() and not g["d_sigmas"][S:].any()
