"""Hand-derived known answers (tests/golden/kat_hand.json: every expected value follows from the reference's formulas by integer or
exact dyadic arithmetic, none was produced by running code) held against THREE implementations alike:

  * `shim`   -- the reference's own kernel source (/root/reference/modules/*.py) executed under oracle/ti_shim: the thing the golden
                fixtures come from, here pinned against something that is not itself (CPU; build container only);
  * `oracle` -- the C restatement (CPU, everywhere);
  * `hip`    -- the product kernels through the C ABI (`-m gpu`).
Plus the notebook regime: the shim-executed march reproduces the sample density a real Taichi run of the reference printed
(notebooks/pipeline.ipynb cells 12-14) within the stated tolerance, and oracle / HIP reproduce the shim's per-ray counts exactly.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "kat_hand.json")))
HAVE_REF = os.path.exists("/root/reference/modules/hash_encoder.py")
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (the shim executes its source)")


@pytest.fixture(scope="module")
def shim():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle"))
    from gen_golden import load_reference
    return load_reference()


def _table_for(entries, n_floats):
    t = np.zeros(n_floats, np.float32)
    for k, (f0, f1) in entries.items():
        t[2 * int(k)], t[2 * int(k) + 1] = f0, f1
    return t


def _hash_cases():
    """(name, x [1,3], table, level, expected [2], exact?)"""
    n_floats = 5710032 * 2
    k = KAT["hash_level0_trilinear"]
    yield "level0_trilinear", np.array([k["x"]], np.float32), _table_for(k["table_entries"], n_floats), k["level"], k["expected"], True
    c = KAT["hash_lattice_corner"]
    for case in c["cases"]:
        yield ("lattice_l%d_%s" % (case["level"], "_".join(map(str, case["cell"]))), np.array([case["x"]], np.float32),
               _table_for({case["entry"]: c["feature"]}, n_floats), case["level"], c["feature"], case["exact"])


def _check_hash(out, level, want, exact, name=""):
    got = out[0, 2 * level:2 * level + 2].tolist()
    if exact:
        assert got == want, (name, got)
    else:               # (scale of this level = f32 exp(): one ulp between implementations, see the case's note)
        np.testing.assert_allclose(got, want, rtol=1e-4, err_msg=name)
    assert np.all(np.delete(np.asarray(out[0]), [2 * level, 2 * level + 1]) == 0), name      # every other level reads zeros


# ------------------------------------------------------------------------------------------------------------ oracle
def test_oracle_scalar_kats(oracle):
    for x, e in KAT["frexp_bit"]["cases"]:
        assert oracle.frexp_bit(x) == e, x
    pts = np.array([c for c, _ in KAT["morton3d"]["cases"]], np.int32)
    assert oracle.morton3d(pts).tolist() == [m for _, m in KAT["morton3d"]["cases"]]
    assert np.array_equal(oracle.morton3d_invert(np.array([m for _, m in KAT["morton3d"]["cases"]], np.int32)), pts)
    pk = KAT["packbits"]
    assert oracle.packbits(np.array(pk["grid"], np.float32), pk["threshold"]).tolist() == [pk["byte"]]


@pytest.mark.parametrize("case", list(_hash_cases()), ids=lambda c: c[0])
def test_oracle_hash_kats(oracle, case):
    name, x, table, level, want, exact = case
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    _check_hash(oracle.hash_fwd_f32(x, table, lv), level, want, exact, name)


def _composite_inputs():
    c = KAT["composite_two_rays"]
    return (np.array(c["sigmas"], np.float32), np.array(c["rgbs"], np.float32), np.array(c["deltas"], np.float32),
            np.array(c["ts"], np.float32), np.array(c["rays_a"], np.int32), c["T_threshold"], c["expected"])


def _check_composite(total, op, dep, rgb, ws, want):
    assert list(map(int, total)) == want["total_samples"]
    assert list(map(float, op)) == want["opacity"] and list(map(float, dep)) == want["depth"]
    assert np.asarray(rgb, np.float32).tolist() == np.asarray(want["rgb"], np.float32).tolist()
    for got, w in zip(ws, want["ws_composited"]):
        if w is not None:                                          # (samples behind the termination point: uninitialised in the reference)
            assert float(got) == w


def test_oracle_composite_kat(oracle):
    s, c, dl, t, ra, thr, want = _composite_inputs()
    total, op, dep, rgb, ws = oracle.composite_train_fwd(s, c, dl, t, ra, thr)
    _check_composite(total, op, dep, rgb, ws, want)


def test_oracle_level_table_is_the_notebooks(oracle):
    nb = KAT["level_table_notebook"]
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    assert lv.total_entries == nb["offset"] and lv.total_entries * lv.n_features == nb["total_hash_size"]
    # per_level_scale = exp(log_b) with log_b = ln(1024 / 16) / 15 in f64 (utils.py:31-39)
    assert abs(np.exp(np.log(1024.0 / 16.0) / 15.0) - nb["per_level_scale"]) < 1e-12


# ------------------------------------------------------------------------------------------------------------ shim (reference source)
@needs_ref
def test_shim_scalar_kats(shim):
    U = shim["utils"]
    import taichi as ti                                            # (the shim; put on sys.path by load_reference)
    for x, e in KAT["frexp_bit"]["cases"]:
        assert int(U.frexp_bit(ti.f32(x))) == e, x
    pts = torch.tensor([c for c, _ in KAT["morton3d"]["cases"]], dtype=torch.int32)
    idx = torch.zeros(len(pts), dtype=torch.int32)
    U.morton3D_kernel(pts, idx)
    assert [int(v) for v in idx] == [m for _, m in KAT["morton3d"]["cases"]]
    pk = KAT["packbits"]
    bits = torch.zeros(1, dtype=torch.uint8)
    U.packbits(torch.tensor(pk["grid"], dtype=torch.float32), pk["threshold"], bits)
    assert int(bits[0]) == pk["byte"]


@needs_ref
@pytest.mark.parametrize("case", list(_hash_cases()), ids=lambda c: c[0])
def test_shim_hash_kats(shim, case):
    name, x, table, level, want, exact = case
    enc = shim["hash_encoder"].HashEncoder(max_params=2**19, levels=16, base_res=16.0, max_res=1024.0, feature_per_level=2)
    out = torch.zeros(1, 32)
    enc._hash_encoder_kernel(torch.from_numpy(x), torch.from_numpy(table), out, enc.hash_map_sizes, enc.offsets, 1)
    _check_hash(out.numpy(), level, want, exact, name)


@needs_ref
def test_shim_composite_kat(shim):
    s, c, dl, t, ra, thr, want = _composite_inputs()
    V = shim["volume_train"]
    n, S = ra.shape[0], s.shape[0]
    total = torch.zeros(n, dtype=torch.int32); op = torch.zeros(n); dep = torch.zeros(n); rgb = torch.zeros(n, 3)
    ws = torch.full((S,), float("nan")); T = torch.zeros(S + 1)
    # (argument order of volume_train.py:7-20)
    V.volume_rendering_kernel(torch.from_numpy(s), torch.from_numpy(c), torch.from_numpy(dl), torch.from_numpy(t), torch.from_numpy(ra), thr,
                              T, total, op, dep, rgb, ws)
    _check_composite(total.numpy(), op.numpy(), dep.numpy(), rgb.numpy(), ws.numpy(), want)


def _regime():
    return np.load(os.path.join(GOLDEN, "ref_notebook_regime.npz"))


def _regime_inputs(g):
    from ngp_hip import synthetic
    n = int(g["n_rays"])
    o, d = synthetic.lego_rays(n, seed=int(g["ray_seed"]))
    bits = synthetic.random_bitfield(1, 128, 0.5, seed=int(g["bits_seed"]))
    noise = np.random.default_rng(int(g["noise_seed"])).random(n, dtype=np.float32)
    return o, d, bits, noise


def test_shim_march_reproduces_the_notebooks_sample_density():
    """The fixture was made by the reference's march kernel under the shim; its samples per ray sit within the stated tolerance of
    what the notebook's real-Taichi run printed (2 055 705 samples / 8192 rays)."""
    g, nb = _regime(), KAT["march_notebook_regime"]
    per_ray = int(g["total"]) / int(g["n_rays"])
    want = nb["samples"] / nb["rays"]
    assert int(g["counts"].sum()) == int(g["total"])
    assert abs(per_ray - want) <= nb["tolerance_relative"] * want, (per_ray, want)
    lt = KAT["level_table_notebook"]
    assert abs(float(g["per_level_scale"]) - lt["per_level_scale"]) < 1e-12 and int(g["total_param_size"]) == lt["total_hash_size"]
    assert int(g["offsets"][-1] + g["hash_map_sizes"][-1]) == lt["offset"]


def test_oracle_reproduces_the_shims_march_in_that_regime(oracle):
    g = _regime()
    o, d, bits, noise = _regime_inputs(g)
    hits = oracle.ray_aabb(o, d, 0.5)
    assert np.array_equal(hits.view(np.uint32), g["hits_t"].view(np.uint32))
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
    assert total == int(g["total"]) and np.array_equal(rays_a[:, 2], g["counts"])
    has = rays_a[:, 2] > 0
    first, last = rays_a[:, 1], rays_a[:, 1] + rays_a[:, 2] - 1
    assert np.array_equal(ts[first[has]].view(np.uint32), g["t_first"][has].view(np.uint32))
    assert np.array_equal(ts[last[has]].view(np.uint32), g["t_last"][has].view(np.uint32))
    assert np.array_equal(deltas[first[has]].view(np.uint32), g["dt_first"][has].view(np.uint32))
    assert float(ts.astype(np.float64).sum()) == float(g["ts_sum"])
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    assert list(lv.offset[:16]) == g["offsets"].tolist() and list(lv.map_size[:16]) == g["hash_map_sizes"].tolist()
    assert lv.begin_fast_hash_level == int(g["begin_fast_hash_level"])


# ------------------------------------------------------------------------------------------------------------ HIP
@pytest.mark.gpu
def test_hip_kats(hip_lib):
    from ngp_hip import ops
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pts = np.array([c for c, _ in KAT["morton3d"]["cases"]], np.int32)
    codes = [m for _, m in KAT["morton3d"]["cases"]]
    assert ops.morton3d(dev(pts)).cpu().tolist() == codes
    assert np.array_equal(ops.morton3d_invert(dev(np.array(codes, np.int32))).cpu().numpy(), pts)
    pk = KAT["packbits"]
    bits = torch.zeros(1, dtype=torch.uint8, device="cuda")
    ops.packbits(dev(np.array(pk["grid"], np.float32)), pk["threshold"], bits)
    assert int(bits[0]) == pk["byte"]
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    for name, x, table, level, want, exact in _hash_cases():
        _check_hash(ops.hash_fwd_f32(dev(x), dev(table), lv).cpu().numpy(), level, want, exact, name)
    s, c, dl, t, ra, thr, want = _composite_inputs()
    total, op, dep, rgb, ws = ops.composite_train_fwd(dev(s), dev(c), dev(dl), dev(t), dev(ra), thr)
    _check_composite(total.cpu().numpy(), op.cpu().numpy(), dep.cpu().numpy(), rgb.cpu().numpy(), ws.cpu().numpy(), want)


@pytest.mark.gpu
def test_hip_reproduces_the_shims_march_in_that_regime(hip_lib):
    """frexp_bit / mip selection / the skip rule have no entry point of their own: they are held through the march (the fixture's
    per-ray counts and first / last samples, made by the reference's source under the shim)."""
    from ngp_hip import ops
    g = _regime()
    o, d, bits, noise = _regime_inputs(g)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    hits = ops.ray_aabb(dev(o), dev(d), 0.5)
    assert np.array_equal(hits.cpu().numpy().view(np.uint32), g["hits_t"].view(np.uint32))
    rays_a, xyzs, dirs, deltas, ts, total = ops.march_train(dev(o), dev(d), hits, dev(bits), dev(noise), 1, 0.5, 0.0, 128, 1024)
    ra, ts, deltas = rays_a.cpu().numpy(), ts.cpu().numpy(), deltas.cpu().numpy()
    assert int(total) == int(g["total"]) and np.array_equal(ra[:, 2], g["counts"])
    has = ra[:, 2] > 0
    first, last = ra[:, 1], ra[:, 1] + ra[:, 2] - 1
    assert np.array_equal(ts[first[has]].view(np.uint32), g["t_first"][has].view(np.uint32))
    assert np.array_equal(ts[last[has]].view(np.uint32), g["t_last"][has].view(np.uint32))
    assert np.array_equal(deltas[first[has]].view(np.uint32), g["dt_first"][has].view(np.uint32))
    assert float(ts[:int(total)].astype(np.float64).sum()) == float(g["ts_sum"])
