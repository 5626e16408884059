"""numpy front-end of the CPU oracle (oracle/ngp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
anything under taichi-nerfs_amd/.  See the header of ngp_oracle.c for what it restates and how it is pinned.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "ngp_oracle.c")
OUT_DIR = os.path.join(_HERE, "_ref")
LIB_PATH = os.path.join(OUT_DIR, "libngp_oracle.so")
CFLAGS = ["-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-fvisibility=hidden"]

sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "taichi-nerfs_amd"))
from ngp_hip.lib import HashLevels  # noqa: E402  (struct layout only; no HIP code is touched)


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdr = os.path.join(os.path.dirname(_HERE), "include", "ngp_hip.h")
    if (not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(hdr)):
        return LIB_PATH
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    res = subprocess.run(["gcc"] + CFLAGS + ["-o", tmp, SRC, "-lm"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("gcc failed building the oracle:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ora_march_train.restype = ctypes.c_int64
        _lib.ora_frexp_bit.restype = ctypes.c_int
        _lib.ora_frexp_bit.argtypes = [ctypes.c_float]
        _lib.ora_hash_levels_init.argtypes = [ctypes.POINTER(HashLevels), ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f(x):
    return ctypes.c_float(float(x))


def make_levels(max_params, levels, base_res, max_res, features):
    lv = HashLevels()
    rc = lib().ora_hash_levels_init(ctypes.byref(lv), float(max_params), int(levels), float(base_res), float(max_res),
                                    int(features))
    assert rc == 0
    return lv


def frexp_bit(x):
    return lib().ora_frexp_bit(float(np.float32(x)))


def ray_aabb(rays_o, rays_d, scale):
    o, d = _f32(rays_o), _f32(rays_d)
    n = o.shape[0]
    hits = np.empty((n, 2), np.float32)
    lib().ora_ray_aabb(_p(o), _p(d), _f(scale), n, _p(hits))
    return hits


def march_train(rays_o, rays_d, hits_t, bitfield, noise, cascades, scale, esf, grid_size, max_samples, count_only=False):
    o, d, h, nz = _f32(rays_o), _f32(rays_d), _f32(hits_t), _f32(noise)
    bits = np.ascontiguousarray(bitfield, dtype=np.uint8)
    n = o.shape[0]
    rays_a = np.empty((n, 3), np.int32)
    L = lib()
    args = [_p(o), _p(d), _p(h), _p(bits), _p(nz), int(cascades), int(grid_size), _f(scale), _f(esf), int(max_samples), n]
    total = L.ora_march_train(*args, _p(rays_a), _p(None), _p(None), _p(None), _p(None))
    if count_only:
        return rays_a, int(total)
    xyzs = np.empty((total, 3), np.float32); dirs = np.empty((total, 3), np.float32)
    deltas = np.empty(total, np.float32); ts = np.empty(total, np.float32)
    # second call writes (the C function sizes nothing itself)
    total2 = L.ora_march_train(*args, _p(rays_a), _p(xyzs), _p(dirs), _p(deltas), _p(ts))
    assert total2 == total
    return rays_a, xyzs, dirs, deltas, ts, int(total)


def march_test(rays_o, rays_d, hits_t, alive, bitfield, cascades, scale, esf, grid_size, max_samples):
    """hits_t is updated in place (must be a contiguous float32 array)."""
    o, d = _f32(rays_o), _f32(rays_d)
    assert hits_t.dtype == np.float32 and hits_t.flags.c_contiguous
    alive = np.ascontiguousarray(alive, dtype=np.int64)
    bits = np.ascontiguousarray(bitfield, dtype=np.uint8)
    n = alive.shape[0]
    ray_indices = np.zeros(n * max_samples, np.int64); valid = np.zeros(n * max_samples, np.uint8)
    deltas = np.zeros(n * max_samples, np.float32); ts = np.zeros(n * max_samples, np.float32)
    counter = np.zeros(n, np.int32)
    lib().ora_march_test(_p(o), _p(d), _p(hits_t), _p(alive), _p(bits), int(cascades), int(grid_size), _f(scale), _f(esf),
                         int(max_samples), n, _p(ray_indices), _p(valid), _p(deltas), _p(ts), _p(counter))
    return ray_indices, valid, deltas, ts, counter


def hash_fwd_f32(xyzs, table, lv):
    x, t = _f32(xyzs), _f32(table).reshape(-1)
    n = x.shape[0]
    out = np.empty((n, lv.n_levels * lv.n_features), np.float32)
    lib().ora_hash_fwd_f32(_p(x), _p(t), ctypes.byref(lv), n, _p(out))
    return out


def hash_corners(xyzs, lv):
    x = _f32(xyzs)
    n = x.shape[0]
    idx = np.empty((n, lv.n_levels, 8), np.uint32); w = np.empty((n, lv.n_levels, 8), np.float32)
    lib().ora_hash_corners(_p(x), ctypes.byref(lv), n, _p(idx), _p(w))
    return idx, w


def hash_bwd_f32(xyzs, dout, lv):
    x, g = _f32(xyzs), _f32(dout)
    dtable = np.zeros(lv.total_entries * lv.n_features, np.float32)
    lib().ora_hash_bwd_f32(_p(x), _p(g), ctypes.byref(lv), x.shape[0], _p(dtable))
    return dtable


def hash_bwd_f32_atomic(xyzs, dout, lv, dtable):
    """cpu_baseline-only throughput variant (omp atomics); accumulates into the caller's dtable."""
    x, g = _f32(xyzs), _f32(dout)
    assert dtable.dtype == np.float32 and dtable.flags.c_contiguous
    lib().ora_hash_bwd_f32_atomic(_p(x), _p(g), ctypes.byref(lv), x.shape[0], _p(dtable))
    return dtable


def hash_fwd_f16(xyzs, table_h, lv):
    x = _f32(xyzs)
    t = np.ascontiguousarray(table_h, dtype=np.float16).view(np.uint16).reshape(-1)
    n = x.shape[0]
    out = np.empty((n, lv.n_levels, lv.n_features), np.uint16)
    lib().ora_hash_fwd_f16(_p(x), _p(t), ctypes.byref(lv), n, _p(out))
    return out.view(np.float16)


def hash_bwd_f16(xyzs, dout_h, lv):
    x = _f32(xyzs)
    g = np.ascontiguousarray(dout_h, dtype=np.float16).view(np.uint16)
    dtable = np.zeros((lv.total_entries, lv.n_features), np.float32)
    lib().ora_hash_bwd_f16(_p(x), _p(g), ctypes.byref(lv), x.shape[0], _p(dtable))
    return dtable


def hash_bwd_f16_serial(xyzs, dout_h, lv):
    """The half2 encoder's explicit backward in the reference's serial order, f16 accumulation: (grad f16 [entries, F],
    contributions per row int32 [entries])."""
    x = _f32(xyzs)
    g = np.ascontiguousarray(dout_h, dtype=np.float16).view(np.uint16)
    dtable = np.zeros((lv.total_entries, lv.n_features), np.uint16)
    count = np.zeros(lv.total_entries, np.int32)
    lib().ora_hash_bwd_f16_serial(_p(x), _p(g), ctypes.byref(lv), x.shape[0], _p(dtable), _p(count))
    return dtable.view(np.float16), count


def sh16_fwd(dirs):
    d = _f32(dirs)
    out = np.empty((d.shape[0], 16), np.float32)
    lib().ora_sh16_fwd(_p(d), d.shape[0], _p(out))
    return out


def sh16_bwd(dirs, dout):
    d, g = _f32(dirs), _f32(dout)
    out = np.empty((d.shape[0], 3), np.float32)
    lib().ora_sh16_bwd(_p(d), _p(g), d.shape[0], _p(out))
    return out


def composite_train_fwd(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    s, c, dl, t = _f32(sigmas), _f32(rgbs), _f32(deltas), _f32(ts)
    ra = np.ascontiguousarray(rays_a, dtype=np.int32)
    n = ra.shape[0]
    total = np.zeros(n, np.int32); op = np.zeros(n, np.float32); dep = np.zeros(n, np.float32)
    rgb = np.zeros((n, 3), np.float32); ws = np.zeros(s.shape[0], np.float32)
    lib().ora_composite_train_fwd(_p(s), _p(c), _p(dl), _p(t), _p(ra), _f(T_threshold), n, _p(total), _p(op), _p(dep), _p(rgb),
                                  _p(ws))
    return total, op, dep, rgb, ws


def composite_train_bwd(g_op, g_dep, g_rgb, g_ws, sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    s, c, dl, t = _f32(sigmas), _f32(rgbs), _f32(deltas), _f32(ts)
    ra = np.ascontiguousarray(rays_a, dtype=np.int32)
    n = ra.shape[0]
    go = _f32(g_op) if g_op is not None else np.zeros(n, np.float32)
    gd = _f32(g_dep) if g_dep is not None else np.zeros(n, np.float32)
    gr = _f32(g_rgb)
    gw = _f32(g_ws) if g_ws is not None else None
    ds = np.zeros(s.shape[0], np.float32); dc = np.zeros((s.shape[0], 3), np.float32)
    lib().ora_composite_train_bwd(_p(go), _p(gd), _p(gr), _p(gw), _p(s), _p(c), _p(dl), _p(t), _p(ra), _f(T_threshold), n,
                                  _p(ds), _p(dc))
    return ds, dc


def composite_test(sigmas, rgbs, deltas, ts, pack_info, alive, T_threshold, opacity, depth, rgb):
    """alive / opacity / depth / rgb updated in place (contiguous arrays of the right dtype)."""
    s, c, dl, t = _f32(sigmas), _f32(rgbs), _f32(deltas), _f32(ts)
    pk = np.ascontiguousarray(pack_info, dtype=np.int64)
    assert alive.dtype == np.int64 and opacity.dtype == np.float32 and depth.dtype == np.float32 and rgb.dtype == np.float32
    lib().ora_composite_test(_p(s), _p(c), _p(dl), _p(t), _p(pk), _p(alive), _f(T_threshold), alive.shape[0], _p(opacity),
                             _p(depth), _p(rgb))


def get_rays(directions, c2w):
    """datasets/ray_utils.py:51-80; c2w [3,4] or [n,3,4]."""
    d, c = _f32(directions), _f32(c2w)
    n = d.shape[0]
    o, r = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    lib().ora_get_rays(_p(d), _p(c), int(c.ndim == 3), n, _p(o), _p(r))
    return o, r


def sample_rays(poses, directions, rays, img_idx, pix_idx):
    """img_idx: int (same_image) or int64 [n]; returns rays_o, rays_d, rgb."""
    ps, d, ry = _f32(poses), _f32(directions), _f32(rays)
    pix = np.ascontiguousarray(pix_idx, dtype=np.int64)
    n = pix.shape[0]
    per = not np.isscalar(img_idx)
    img = np.ascontiguousarray(img_idx, dtype=np.int64) if per else None
    o, r, c = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    lib().ora_sample_rays(_p(ps), _p(d), _p(ry), ry.shape[-1], ctypes.c_longlong(ry.shape[1]), _p(img),
                          ctypes.c_longlong(0 if per else int(img_idx)), _p(pix), n, _p(o), _p(r), _p(c))
    return o, r, c


def morton3d(coords):
    c = np.ascontiguousarray(coords, dtype=np.int32)
    out = np.empty(c.shape[0], np.int32)
    lib().ora_morton3d(_p(c), c.shape[0], _p(out))
    return out


def morton3d_invert(indices):
    i = np.ascontiguousarray(indices, dtype=np.int32)
    out = np.empty((i.shape[0], 3), np.int32)
    lib().ora_morton3d_invert(_p(i), i.shape[0], _p(out))
    return out


def packbits(grid, thr):
    g = _f32(grid).reshape(-1)
    out = np.empty(g.shape[0] // 8, np.uint8)
    lib().ora_packbits(_p(g), _f(thr), out.shape[0], _p(out))
    return out


def distortion_fwd(ws, deltas, ts, rays_a):
    w, dl, t = _f32(ws), _f32(deltas), _f32(ts)
    ra = np.ascontiguousarray(rays_a, dtype=np.int32)
    loss = np.zeros(ra.shape[0], np.float32)
    wi = np.zeros_like(w); wti = np.zeros_like(w)
    lib().ora_distortion_fwd(_p(w), _p(dl), _p(t), _p(ra), ra.shape[0], _p(loss), _p(wi), _p(wti))
    return loss, wi, wti


def distortion_bwd(dL_dloss, deltas, ws, ts, ws_inc, wts_inc, rays_a):
    g, dl, w, t = _f32(dL_dloss), _f32(deltas), _f32(ws), _f32(ts)
    ra = np.ascontiguousarray(rays_a, dtype=np.int32)
    out = np.zeros_like(w)
    lib().ora_distortion_bwd(_p(g), _p(dl), _p(w), _p(t), _p(_f32(ws_inc)), _p(_f32(wts_inc)), _p(ra), ra.shape[0], _p(out))
    return out


def f32_to_f16_bits(a):
    x = _f32(a).reshape(-1)
    out = np.empty(x.shape[0], np.uint16)
    lib().ora_f32_to_f16(_p(x), x.shape[0], _p(out))
    return out.reshape(np.shape(a))
