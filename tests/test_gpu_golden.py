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


def test_hash_f16_golden(hip_lib):
    g = G("ref_hash_f16.npz")
    lv = ops.make_levels(2**19, 16, 16.0, 1024.0, 2)
    for l in range(16):
        lv.scale[l] = float(g["scale_used"][l])
    table_h = golden_table(int(g["total_entries"]) * 2, -0.1, 0.1).astype(np.float16).reshape(-1, 2)
    out = ops.hash_fwd_f16(dev(g["xyzs"]), dev(table_h), lv).cpu().numpy()
    diff = np.abs(out.astype(np.float32) - g["out"].astype(np.float32))
    assert (diff == 0).mean() > 0.97 and diff.max() <= 2.5e-4
    grad = torch.zeros(table_h.shape[0], 2, device="cuda", dtype=torch.float16)
    ops.hash_bwd_f16(dev(g["xyzs"]), dev(g["dout"]), lv, grad)
    np.testing.assert_allclose(grad.float().cpu().numpy()[g["grad_rows"]], g["grad_vals"].astype(np.float32), rtol=2e-2, atol=2e-5)


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
