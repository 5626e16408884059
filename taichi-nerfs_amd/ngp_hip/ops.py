"""Tensor-level launchers over the C ABI: allocate outputs with torch, pass raw device pointers + the current
torch HIP stream.  No autograd here (that lives in modules/), no CPU path: every function requires device
tensors and raises otherwise."""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib_mod
from .lib import HashLevels, check


def _lib():
    return _lib_mod.load()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream on the current device as a void* (every launch goes there).  torch.cuda.current_stream() builds
    a Stream object per call (~10 us, nine calls per training step of the operator path: profiles/r04_train_py_cprofile.txt);
    the raw getter is the same handle without the object."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _touched(*tensors):
    """A kernel wrote these tensors IN PLACE through their raw pointers: move their torch version counters like any in-place
    torch op would, so version-keyed caches (the trainer's coarse occupancy table) and autograd's saved-tensor checks see it."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _dev(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: libngp_hip has no CPU path (got device %s)" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def make_levels(max_params, levels, base_res, max_res, features):
    """ngp_hash_levels table (host struct) -- HashEncoder.__init__ arithmetic (hash_encoder.py:183-205)."""
    lv = HashLevels()
    check(_lib().ngp_hash_levels_init(ctypes.byref(lv), float(max_params), int(levels), float(base_res), float(max_res),
                                      int(features)), "ngp_hash_levels_init")
    return lv


# ---------------------------------------------------------------------------------------------------- a-1
def ray_aabb(rays_o, rays_d, scale):
    _dev(rays_o, torch.float32, "rays_o"); _dev(rays_d, torch.float32, "rays_d")
    n = rays_o.shape[0]
    hits_t = torch.empty(n, 2, device=rays_o.device, dtype=torch.float32)
    check(_lib().ngp_ray_aabb(_ptr(rays_o), _ptr(rays_d), float(scale), n, _ptr(hits_t), _stream()), "ngp_ray_aabb")
    return hits_t


# ---------------------------------------------------------------------------------------------------- a-2
class MarchArena:
    """Per-device staging rows for the training march: [n_rays * max_samples] (t, dt) pairs.

    Worst case like the reference's N*1024 torch.empty buffers (ray_march.py:144-168) but allocated once and
    reused; only the rows/entries actually emitted are ever touched (288 GB of HBM makes the reservation free)."""
    _cache = {}

    @classmethod
    def get(cls, device, n_rays, max_samples):
        key = (device.index if device.index is not None else torch.cuda.current_device())
        need = n_rays * max_samples
        cur = cls._cache.get(key)
        if cur is None or cur.shape[0] < need:
            cur = torch.empty(need, 2, device=device, dtype=torch.float32)
            cls._cache[key] = cur
        return cur


def coarse_bitfield(density_bitfield, cascades, grid_size, out=None):
    """One bit per 8^3-cell block (any cell occupied) -- the shortcut table of ngp_march_train_count_ex."""
    words = int(cascades) * int(grid_size)**3 // 512 // 32
    if out is None:
        out = torch.empty(words, device=density_bitfield.device, dtype=torch.int32)
    check(_lib().ngp_bitfield_coarsen(_ptr(density_bitfield), int(cascades), int(grid_size), _ptr(out), _stream()),
          "ngp_bitfield_coarsen")
    return out


def march_train(rays_o, rays_d, hits_t, density_bitfield, noise, cascades, scale, exp_step_factor, grid_size, max_samples):
    """count -> scan -> (one D2H read of the total, like ray_march.py:187-192) -> write."""
    _dev(rays_o, torch.float32, "rays_o"); _dev(rays_d, torch.float32, "rays_d"); _dev(hits_t, torch.float32, "hits_t")
    _dev(density_bitfield, torch.uint8, "density_bitfield"); _dev(noise, torch.float32, "noise")
    n = rays_o.shape[0]
    dev = rays_o.device
    L = _lib()
    stage = MarchArena.get(dev, n, int(max_samples))
    counts = torch.empty(n, device=dev, dtype=torch.int32)
    rays_a = torch.empty(n, 3, device=dev, dtype=torch.int32)
    total = torch.zeros(1, device=dev, dtype=torch.int32)
    st = _stream()
    coarse = coarse_bitfield(density_bitfield, cascades, grid_size)
    check(L.ngp_march_train_count_ex(_ptr(rays_o), _ptr(rays_d), _ptr(hits_t), _ptr(density_bitfield), _ptr(coarse), _ptr(noise),
                                     int(cascades), int(grid_size), float(scale), float(exp_step_factor), int(max_samples), n,
                                     _ptr(stage), _ptr(counts), st), "ngp_march_train_count_ex")
    check(L.ngp_march_train_scan(_ptr(counts), n, _ptr(rays_a), _ptr(total), st), "ngp_march_train_scan")
    S = int(total.item())
    xyzs = torch.empty(S, 3, device=dev, dtype=torch.float32)
    dirs = torch.empty(S, 3, device=dev, dtype=torch.float32)
    deltas = torch.empty(S, device=dev, dtype=torch.float32)
    ts = torch.empty(S, device=dev, dtype=torch.float32)
    if S > 0:
        check(L.ngp_march_train_write(_ptr(rays_o), _ptr(rays_d), _ptr(rays_a), _ptr(stage), int(max_samples), n,
                                      _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(ts), st), "ngp_march_train_write")
    return rays_a, xyzs, dirs, deltas, ts, total[0]


def march_train_fused(rays_o, rays_d, hits_t, density_bitfield, noise, cascades, scale, exp_step_factor, grid_size, max_samples,
                      capacity=None, seed=None, shape=None):
    """ngp_march_train_fused: the same samples per ray as march_train() in ONE launch, the rays packed in block-completion order
    (rays_a[r] = (r, start, count)); outputs are sized for `capacity` samples (default n * max_samples), the first `total` valid --
    with a smaller capacity the kernel drops what does not fit (compare the returned total with it).  hits_t may be None (slab
    test inline).  noise=None + seed: the jitter is drawn in the kernel, ray r gets rng_uniform(seed, r).  shape = (waves per
    block in {4, 8, 16}, bytes of idle dynamic LDS per block): ngp_march_train_fused_shaped (full-capacity outputs only)."""
    _dev(rays_o, torch.float32, "rays_o"); _dev(rays_d, torch.float32, "rays_d")
    _dev(density_bitfield, torch.uint8, "density_bitfield")
    if noise is None:
        if seed is None:
            raise ValueError("march_train_fused needs a noise vector or a seed")
    else:
        _dev(noise, torch.float32, "noise")
    n = rays_o.shape[0]
    dev = rays_o.device
    L = _lib()
    stage = MarchArena.get(dev, n, int(max_samples))
    cap = n * int(max_samples) if capacity is None else int(capacity)
    rays_a = torch.empty(n, 3, device=dev, dtype=torch.int32)
    total = torch.zeros(1, device=dev, dtype=torch.int32)
    ctr = torch.zeros(2, device=dev, dtype=torch.int32)
    xyzs, dirs = torch.empty(cap, 3, device=dev, dtype=torch.float32), torch.empty(cap, 3, device=dev, dtype=torch.float32)
    deltas, ts = torch.empty(cap, device=dev, dtype=torch.float32), torch.empty(cap, device=dev, dtype=torch.float32)
    coarse = coarse_bitfield(density_bitfield, cascades, grid_size)
    if shape is not None:
        if cap < n * int(max_samples):
            raise ValueError("the shaped form writes into arrays of n * max_samples rows")
        check(L.ngp_march_train_fused_shaped(_ptr(rays_o), _ptr(rays_d), _ptr(hits_t), _ptr(density_bitfield), _ptr(coarse), _ptr(noise),
                                             int(seed or 0), int(cascades), int(grid_size), float(scale), float(exp_step_factor),
                                             int(max_samples), n, int(shape[0]), int(shape[1]), _ptr(stage), _ptr(ctr), _ptr(rays_a),
                                             _ptr(total), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(ts), _stream()),
              "ngp_march_train_fused_shaped")
    elif noise is None:
        if cap < n * int(max_samples):
            raise ValueError("the seeded form writes into arrays of n * max_samples rows")
        check(L.ngp_march_train_fused_rng(_ptr(rays_o), _ptr(rays_d), _ptr(hits_t), _ptr(density_bitfield), _ptr(coarse), int(seed),
                                          int(cascades), int(grid_size), float(scale), float(exp_step_factor), int(max_samples), n,
                                          _ptr(stage), _ptr(ctr), _ptr(rays_a), _ptr(total), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(ts),
                                          _stream()), "ngp_march_train_fused_rng")
    else:
        check(L.ngp_march_train_fused_cap(_ptr(rays_o), _ptr(rays_d), _ptr(hits_t), _ptr(density_bitfield), _ptr(coarse), _ptr(noise),
                                          int(cascades), int(grid_size), float(scale), float(exp_step_factor), int(max_samples), n, cap,
                                          _ptr(stage), _ptr(ctr), _ptr(rays_a), _ptr(total), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(ts),
                                          _stream()), "ngp_march_train_fused_cap")
    return rays_a, xyzs, dirs, deltas, ts, total[0], ctr


def rng_uniform(seed, n, device="cuda"):
    """[n] float32: rng_uniform(seed, i) -- the jitter ngp_march_train_fused_rng(seed) gives ray i."""
    out = torch.empty(n, device=device, dtype=torch.float32)
    check(_lib().ngp_rng_uniform(int(seed), int(n), _ptr(out), _stream()), "ngp_rng_uniform")
    return out


# ---------------------------------------------------------------------------------------------------- a-3
def march_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor, grid_size,
               max_samples):
    _dev(rays_o, torch.float32, "rays_o"); _dev(rays_d, torch.float32, "rays_d"); _dev(hits_t, torch.float32, "hits_t")
    _dev(alive_indices, torch.int64, "alive_indices"); _dev(density_bitfield, torch.uint8, "density_bitfield")
    n = alive_indices.shape[0]
    dev = rays_o.device
    ray_indices = torch.empty(n * max_samples, device=dev, dtype=torch.int64)
    valid_mask = torch.zeros(n * max_samples, device=dev, dtype=torch.uint8)
    deltas = torch.empty(n * max_samples, device=dev, dtype=torch.float32)
    ts = torch.empty(n * max_samples, device=dev, dtype=torch.float32)
    samples_counter = torch.empty(n, device=dev, dtype=torch.int32)
    check(_lib().ngp_march_test(_ptr(rays_o), _ptr(rays_d), _ptr(hits_t), _ptr(alive_indices), _ptr(density_bitfield),
                                int(cascades), int(grid_size), float(scale), float(exp_step_factor), int(max_samples), n,
                                _ptr(ray_indices), _ptr(valid_mask), _ptr(deltas), _ptr(ts), _ptr(samples_counter), _stream()),
          "ngp_march_test")
    _touched(hits_t)                                                  # resume state, advanced in place (ray_march.py:262-268)
    return ray_indices, valid_mask, deltas, ts, samples_counter


# ---------------------------------------------------------------------------------------------------- a-4/5
def hash_fwd_f32(xyzs, table, lv):
    _dev(xyzs, torch.float32, "xyzs"); _dev(table, torch.float32, "hash_table")
    n = xyzs.shape[0]
    out = torch.empty(n, lv.n_levels * lv.n_features, device=xyzs.device, dtype=torch.float32)
    check(_lib().ngp_hash_fwd_f32(_ptr(xyzs), _ptr(table), ctypes.byref(lv), n, _ptr(out), _stream()), "ngp_hash_fwd_f32")
    return out


def cast_bf16(src, dst=None):
    """dst = bf16(src), round-to-nearest-even (ngp_cast_f32_bf16); numel % 4 == 0."""
    _dev(src, torch.float32, "src")
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _dev(dst, torch.bfloat16, "dst")
    check(_lib().ngp_cast_f32_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()), "ngp_cast_f32_bf16")
    return dst


def hash_fwd_bf16(xyzs, table_bf16, lv):
    """bf16-stored table (F = 2), f32 interpolation and output [n, L*2]."""
    _dev(xyzs, torch.float32, "xyzs"); _dev(table_bf16, torch.bfloat16, "hash_table(bf16)")
    n = xyzs.shape[0]
    out = torch.empty(n, lv.n_levels * lv.n_features, device=xyzs.device, dtype=torch.float32)
    check(_lib().ngp_hash_fwd_bf16_ex(_ptr(xyzs), _ptr(table_bf16), ctypes.byref(lv), n, _ptr(None), 0, 0.0, 1.0, 0, _ptr(out),
                                      _stream()), "ngp_hash_fwd_bf16_ex")
    return out


SLICED_MIN_SAMPLES = 65536      # below this the float-atomic kernel wins (the sliced launch has a ~60 us floor: 1 k slice owners)


def hash_bwd_f32(xyzs, dout, lv, dtable):
    """dtable += scatter-add of dout (hash_encoder.py:269, the Taichi-autodiff backward).  Large batches take the LDS-sliced
    formulation (no global float atomics, f64 accumulation; 3x faster at training batch sizes), small ones -- or level tables
    it cannot express (F != 2, a level of more than 64 slices) -- the float-atomic kernel.  NGP_HASH_BWD=atomic forces the latter."""
    _dev(xyzs, torch.float32, "xyzs"); _dev(dout, torch.float32, "dout"); _dev(dtable, torch.float32, "dtable")
    n = xyzs.shape[0]
    if n >= SLICED_MIN_SAMPLES and os.environ.get("NGP_HASH_BWD", "sliced") != "atomic" and lv.n_features == 2:
        ws = sliced_workspace(lv, n, xyzs.device)
        rc = _lib().ngp_hash_bwd_f32_sliced(_ptr(xyzs), _ptr(dout), ctypes.byref(lv), n, _ptr(None), _ptr(None), 0, 0.0, 1.0, 0,
                                            _ptr(dtable), _ptr(None), _ptr(ws), ws.numel(), _stream())
        if rc != -2:
            check(rc, "ngp_hash_bwd_f32_sliced")
            _touched(dtable)
            return dtable
    check(_lib().ngp_hash_bwd_f32(_ptr(xyzs), _ptr(dout), ctypes.byref(lv), n, _ptr(dtable), _stream()),
          "ngp_hash_bwd_f32")
    _touched(dtable)
    return dtable


_sliced_ws = {}


def sliced_workspace(lv, n_max, device):
    """Scratch of the LDS-sliced scatter-add for buffers of n_max samples: one per (device, stream) -- the prepass's bitmaps and the
    persistent workgroups' queue heads live in it, so two backward passes in flight on different streams must not share one
    (on one stream the launches serialise and the buffer is reused).  Grown, never shrunk; a buffer that is replaced is
    handed back to torch's caching allocator, which keeps it away from other streams until this stream's pending work is done."""
    need = int(_lib().ngp_hash_bwd_sliced_workspace(ctypes.byref(lv), int(n_max)))
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, torch.cuda.current_stream(device).cuda_stream)
    ws = _sliced_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _sliced_ws[key] = torch.empty(need, device=device, dtype=torch.uint8)
    return ws


def hash_bwd_f32_sliced(xyzs, dout, lv, dtable, live_idx=None, n_dev=None):
    """dtable += scatter-add of dout (natural [n, L*2] layout), LDS-sliced formulation; xyzs in [0,1].  Raises if the level table
    does not fit (use hash_bwd_f32 then)."""
    _dev(xyzs, torch.float32, "xyzs"); _dev(dout, torch.float32, "dout"); _dev(dtable, torch.float32, "dtable")
    n = dout.shape[0]
    ws = sliced_workspace(lv, n, xyzs.device)
    check(_lib().ngp_hash_bwd_f32_sliced(_ptr(xyzs), _ptr(dout), ctypes.byref(lv), n, _ptr(n_dev), _ptr(live_idx), 0, 0.0, 1.0, 0,
                                         _ptr(dtable), _ptr(None), _ptr(ws), ws.numel(), _stream()), "ngp_hash_bwd_f32_sliced")
    _touched(dtable)
    return dtable


def hash_fwd_f16(xyzs, table_h, lv):
    _dev(xyzs, torch.float32, "xyzs"); _dev(table_h, torch.float16, "hash_table(f16)")
    n = xyzs.shape[0]
    out = torch.empty(n, lv.n_levels, lv.n_features, device=xyzs.device, dtype=torch.float16)
    check(_lib().ngp_hash_fwd_f16(_ptr(xyzs), _ptr(table_h), ctypes.byref(lv), n, _ptr(out), _stream()), "ngp_hash_fwd_f16")
    return out


def hash_bwd_f16(xyzs, dout_h, lv, dtable_h):
    """The half2 encoder's scatter-add (hash_encoder_half.py:163-213).  Like hash_bwd_f32: large batches take the LDS-sliced
    form (exact sum of the fp16 contributions, rounded once), small ones / unsupported tables the packed-f16-atomic kernel."""
    _dev(xyzs, torch.float32, "xyzs"); _dev(dout_h, torch.float16, "dout"); _dev(dtable_h, torch.float16, "dtable")
    n = xyzs.shape[0]
    if n >= SLICED_MIN_SAMPLES and os.environ.get("NGP_HASH_BWD", "sliced") != "atomic" and lv.n_features == 2:
        ws = sliced_workspace(lv, n, xyzs.device)
        L = _lib()
        rc = L.ngp_hash_bwd_sliced_prep(_ptr(xyzs), ctypes.byref(lv), n, _ptr(None), _ptr(None), 0, 0.0, 1.0, _ptr(ws), ws.numel(), _stream())
        if rc != -2:
            check(rc, "ngp_hash_bwd_sliced_prep")
            dout = dout_h.float().reshape(n, -1).contiguous()          # fp16 -> fp32 is exact; the kernel rounds back (a no-op)
            check(L.ngp_hash_bwd_sliced_main_f16(_ptr(dout), ctypes.byref(lv), n, _ptr(None), 0, _ptr(dtable_h), _ptr(None), _ptr(ws),
                                                 ws.numel(), _stream()), "ngp_hash_bwd_sliced_main_f16")
            _touched(dtable_h)
            return dtable_h
    check(_lib().ngp_hash_bwd_f16(_ptr(xyzs), _ptr(dout_h), ctypes.byref(lv), xyzs.shape[0], _ptr(dtable_h), _stream()),
          "ngp_hash_bwd_f16")
    _touched(dtable_h)
    return dtable_h


def hash_bwd_f16_sliced(xyzs, dout, lv, dtable_h, live_idx=None, n_dev=None):
    """dtable_h (fp16 [entries, 2]) += the half2 encoder's scatter-add of dout (fp32 [n, L*2], rounded to fp16 like the encoder's
    output gradient) in the LDS-sliced formulation: prepass + ngp_hash_bwd_sliced_main_f16; xyzs in [0,1]."""
    _dev(xyzs, torch.float32, "xyzs"); _dev(dout, torch.float32, "dout"); _dev(dtable_h, torch.float16, "dtable")
    n = dout.shape[0]
    ws = sliced_workspace(lv, n, xyzs.device)
    L = _lib()
    check(L.ngp_hash_bwd_sliced_prep(_ptr(xyzs), ctypes.byref(lv), n, _ptr(n_dev), _ptr(live_idx), 0, 0.0, 1.0, _ptr(ws), ws.numel(),
                                     _stream()), "ngp_hash_bwd_sliced_prep")
    check(L.ngp_hash_bwd_sliced_main_f16(_ptr(dout), ctypes.byref(lv), n, _ptr(n_dev), 0, _ptr(dtable_h), _ptr(None), _ptr(ws), ws.numel(),
                                         _stream()), "ngp_hash_bwd_sliced_main_f16")
    _touched(dtable_h)
    return dtable_h


# ---------------------------------------------------------------------------------------------------- a-6
def sh16_fwd(dirs):
    _dev(dirs, torch.float32, "dirs")
    out = torch.empty(dirs.shape[0], 16, device=dirs.device, dtype=torch.float32)
    check(_lib().ngp_sh16_fwd(_ptr(dirs), dirs.shape[0], _ptr(out), _stream()), "ngp_sh16_fwd")
    return out


def sh16_bwd(dirs, dout):
    _dev(dirs, torch.float32, "dirs"); _dev(dout, torch.float32, "dout")
    ddirs = torch.empty_like(dirs)
    check(_lib().ngp_sh16_bwd(_ptr(dirs), _ptr(dout), dirs.shape[0], _ptr(ddirs), _stream()), "ngp_sh16_bwd")
    return ddirs


# ---------------------------------------------------------------------------------------------------- a-7
def _rgb_kind(rgbs):
    if rgbs.dtype == torch.float16:
        return 1
    if rgbs.dtype == torch.float32:
        return 0
    raise TypeError("rgbs must be float16 or float32, got %s" % rgbs.dtype)


def composite_train_fwd(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    _dev(sigmas, torch.float32, "sigmas"); _dev(rgbs, None, "rgbs"); _dev(deltas, torch.float32, "deltas")
    _dev(ts, torch.float32, "ts"); _dev(rays_a, torch.int32, "rays_a")
    n = rays_a.shape[0]
    dev = rays_a.device
    total_samples = torch.empty(n, device=dev, dtype=torch.int32)
    opacity = torch.empty(n, device=dev, dtype=torch.float32)
    depth = torch.empty(n, device=dev, dtype=torch.float32)
    rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
    ws = torch.empty_like(sigmas)
    check(_lib().ngp_composite_train_fwd(_ptr(sigmas), _ptr(rgbs), _rgb_kind(rgbs), _ptr(deltas), _ptr(ts), _ptr(rays_a),
                                         float(T_threshold), n, _ptr(total_samples), _ptr(opacity), _ptr(depth), _ptr(rgb),
                                         _ptr(ws), _stream()), "ngp_composite_train_fwd")
    return total_samples, opacity, depth, rgb, ws


def composite_train_bwd(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws,
                        T_threshold):
    n = rays_a.shape[0]
    for name, t in (("dL_dopacity", dL_dopacity), ("dL_ddepth", dL_ddepth), ("dL_drgb", dL_drgb), ("dL_dws", dL_dws)):
        if t is not None:
            _dev(t, torch.float32, name)
    d_sigmas = torch.empty_like(sigmas)
    d_rgbs = torch.empty_like(rgbs)
    check(_lib().ngp_composite_train_bwd(_ptr(dL_dopacity), _ptr(dL_ddepth), _ptr(dL_drgb), _ptr(dL_dws), _ptr(sigmas),
                                         _ptr(rgbs), _rgb_kind(rgbs), _ptr(deltas), _ptr(ts), _ptr(rays_a), _ptr(opacity),
                                         _ptr(depth), _ptr(rgb), _ptr(ws), float(T_threshold), n, _ptr(d_sigmas), _ptr(d_rgbs),
                                         _stream()), "ngp_composite_train_bwd")
    return d_sigmas, d_rgbs


# ---------------------------------------------------------------------------------------------------- a-8
def composite_test(sigmas, rgbs, deltas, ts, pack_info, alive_indices, T_threshold, opacity, depth, rgb):
    _dev(sigmas, torch.float32, "sigmas"); _dev(rgbs, None, "rgbs"); _dev(deltas, torch.float32, "deltas")
    _dev(ts, torch.float32, "ts"); _dev(pack_info, torch.int64, "pack_info"); _dev(alive_indices, torch.int64, "alive_indices")
    _dev(opacity, torch.float32, "opacity"); _dev(depth, torch.float32, "depth"); _dev(rgb, torch.float32, "rgb")
    check(_lib().ngp_composite_test(_ptr(sigmas), _ptr(rgbs), _rgb_kind(rgbs), _ptr(deltas), _ptr(ts), _ptr(pack_info),
                                    _ptr(alive_indices), float(T_threshold), alive_indices.shape[0], _ptr(opacity), _ptr(depth),
                                    _ptr(rgb), _stream()), "ngp_composite_test")
    _touched(opacity, depth, rgb, alive_indices)


# ---------------------------------------------------------------------------------------------------- a-9
MLP_N_WEIGHTS = 2048 + 1024 + 2048 + 4096 + 192
MLP_SPLITS = (2048, 1024, 2048, 4096, 192)
MLP_SHAPES = ((64, 32), (16, 64), (64, 32), (64, 64), (3, 64))


def mlp_pack(weights):
    """weights: the five fp32 nn.Linear weight tensors (W1 [64,32], W2 [16,64], W3 [64,32], W4 [64,64], W5 [3,64])
    -> fp16 MFMA-fragment image consumed by mlp_fwd / mlp_bwd."""
    ws = []
    for w, shape in zip(weights, MLP_SHAPES):
        if tuple(w.shape) != shape:
            raise ValueError("fused MLP needs the default architecture; got weight shape %s, want %s" % (tuple(w.shape), shape))
        ws.append(_dev(w.detach().contiguous(), torch.float32, "mlp weight"))
    wpack = torch.empty(_lib().ngp_mlp_wpack_halfs(), device=ws[0].device, dtype=torch.float16)
    check(_lib().ngp_mlp_pack(*[_ptr(w) for w in ws], 0, _ptr(wpack), _stream()), "ngp_mlp_pack")
    return wpack


def mlp_fwd(enc, dirs, wpack):
    """enc [n,32] f32, dirs [n,3] f32 (raw directions) -> (sigmas [n] f32, rgbs [n,3] f16)."""
    _dev(enc, torch.float32, "enc"); _dev(dirs, torch.float32, "dirs"); _dev(wpack, torch.float16, "wpack")
    n = enc.shape[0]
    sigmas = torch.empty(n, device=enc.device, dtype=torch.float32)
    rgbs = torch.empty(n, 3, device=enc.device, dtype=torch.float16)
    check(_lib().ngp_mlp_fwd(_ptr(enc), _ptr(dirs), _ptr(wpack), n, _ptr(sigmas), _ptr(rgbs), _stream()), "ngp_mlp_fwd")
    return sigmas, rgbs


def mlp_density(enc, wpack):
    """density head only: enc [n,32] f32 -> sigmas [n] f32."""
    _dev(enc, torch.float32, "enc"); _dev(wpack, torch.float16, "wpack")
    n = enc.shape[0]
    sigmas = torch.empty(n, device=enc.device, dtype=torch.float32)
    check(_lib().ngp_mlp_fwd(_ptr(enc), _ptr(None), _ptr(wpack), n, _ptr(sigmas), _ptr(None), _stream()), "ngp_mlp_fwd")
    return sigmas


def mlp_bwd(enc, dirs, wpack, dsigmas, drgbs):
    """-> (d_enc [n,32] f32, dW [9408] f32 flat = W1|W2|W3|W4|W5)."""
    _dev(enc, torch.float32, "enc"); _dev(dirs, torch.float32, "dirs"); _dev(wpack, torch.float16, "wpack")
    _dev(dsigmas, torch.float32, "dsigmas"); _dev(drgbs, torch.float16, "drgbs")
    n = enc.shape[0]
    d_enc = torch.empty_like(enc)
    dW = torch.zeros(MLP_N_WEIGHTS, device=enc.device, dtype=torch.float32)
    check(_lib().ngp_mlp_bwd(_ptr(enc), _ptr(dirs), _ptr(wpack), _ptr(dsigmas), _ptr(drgbs), n, _ptr(d_enc), _ptr(dW), _stream()),
          "ngp_mlp_bwd")
    return d_enc, dW


# ---------------------------------------------------------------------------------------------------- a-10
def morton3d(coords):
    _dev(coords, torch.int32, "coords")
    out = torch.empty(coords.shape[0], device=coords.device, dtype=torch.int32)
    check(_lib().ngp_morton3d(_ptr(coords), coords.shape[0], _ptr(out), _stream()), "ngp_morton3d")
    return out


def morton3d_invert(indices):
    _dev(indices, torch.int32, "indices")
    out = torch.empty(indices.shape[0], 3, device=indices.device, dtype=torch.int32)
    check(_lib().ngp_morton3d_invert(_ptr(indices), indices.shape[0], _ptr(out), _stream()), "ngp_morton3d_invert")
    return out


def packbits(density_grid, threshold, density_bitfield):
    _dev(density_grid, torch.float32, "density_grid"); _dev(density_bitfield, torch.uint8, "density_bitfield")
    n_bytes = density_bitfield.shape[0]
    if density_grid.numel() != 8 * n_bytes:
        raise ValueError("density_grid must hold 8 floats per bitfield byte")
    check(_lib().ngp_packbits(_ptr(density_grid), float(threshold), n_bytes, _ptr(density_bitfield), _stream()), "ngp_packbits")
    _touched(density_bitfield)
    return density_bitfield


# ---------------------------------------------------------------------------------------------------- f-1
def distortion_fwd(ws, deltas, ts, rays_a):
    _dev(ws, torch.float32, "ws"); _dev(deltas, torch.float32, "deltas"); _dev(ts, torch.float32, "ts")
    _dev(rays_a, torch.int32, "rays_a")
    n = rays_a.shape[0]
    loss = torch.zeros(n, device=ws.device, dtype=torch.float32)
    ws_inc = torch.empty_like(ws)
    wts_inc = torch.empty_like(ws)
    check(_lib().ngp_distortion_fwd(_ptr(ws), _ptr(deltas), _ptr(ts), _ptr(rays_a), n, _ptr(loss), _ptr(ws_inc), _ptr(wts_inc),
                                    _stream()), "ngp_distortion_fwd")
    return loss, ws_inc, wts_inc


def distortion_bwd(dL_dloss, ws, deltas, ts, ws_inc, wts_inc, rays_a):
    _dev(dL_dloss, torch.float32, "dL_dloss")
    dL_dws = torch.zeros_like(ws)
    check(_lib().ngp_distortion_bwd(_ptr(dL_dloss), _ptr(ws), _ptr(deltas), _ptr(ts), _ptr(ws_inc), _ptr(wts_inc), _ptr(rays_a),
                                    rays_a.shape[0], _ptr(dL_dws), _stream()), "ngp_distortion_bwd")
    return dL_dws


def levels_to_numpy(lv):
    """(scale, resolution, map_size, offset) as numpy arrays -- for tests and for the module buffers."""
    L = lv.n_levels
    return (np.array(lv.scale[:L], dtype=np.float32), np.array(lv.resolution[:L], dtype=np.uint32),
            np.array(lv.map_size[:L], dtype=np.uint32), np.array(lv.offset[:L], dtype=np.uint32))
