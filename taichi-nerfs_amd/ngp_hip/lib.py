"""ctypes binding of libngp_hip.so (C ABI declared in include/ngp_hip.h).

This is the binding a maintainer of the reference would add in place of Taichi's torch interop: every
`@ti.kernel` launch in the reference's modules/*.py becomes one call through this table with
`tensor.data_ptr()` + the current torch stream.  There is NO fallback: if the shared library is missing or
fails to load, `load()` raises -- the product path never routes through a CPU implementation.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(PKG_ROOT, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_ROOT), "include")
LIB_PATH = os.path.join(CSRC, "libngp_hip.so")
SOURCES = ["march.hip", "hash_grid.hip", "hash_bwd_lds.hip", "composite.hip", "sh_grid.hip", "mlp.hip", "optim.hip", "distortion.hip", "occupancy.hip", "rays.hip",
           "render.hip", "exchange.hip"]
HEADERS = ["ngp_device.h", "hash_common.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
FLAGS_STAMP = LIB_PATH + ".flags"      # the flags the in-tree .so was built with: a flag change rebuilds

NGP_MAX_LEVELS = 16


class HashLevels(ctypes.Structure):
    """Mirror of `ngp_hash_levels` (include/ngp_hip.h)."""
    _fields_ = [
        ("n_levels", ctypes.c_int32),
        ("n_features", ctypes.c_int32),
        ("begin_fast_hash_level", ctypes.c_int32),
        ("total_entries", ctypes.c_int32),
        ("scale", ctypes.c_float * NGP_MAX_LEVELS),
        ("resolution", ctypes.c_uint32 * NGP_MAX_LEVELS),
        ("map_size", ctypes.c_uint32 * NGP_MAX_LEVELS),
        ("offset", ctypes.c_uint32 * NGP_MAX_LEVELS),
        ("bwd_plan", ctypes.c_uint32),          # NGP_BWD_PLAN_* bits (task plan of the LDS-sliced scatter-add over this table)
    ]

    def with_plan(self, bits):
        """A copy of this level table whose LDS-sliced scatter-add runs on the task plan `bits` (BWD_PLAN_*); copies are cached."""
        bits = int(bits)
        if bits == self.bwd_plan:
            return self
        cache = self.__dict__.setdefault("_plans", {})
        lv = cache.get(bits)
        if lv is None:
            lv = HashLevels()
            ctypes.memmove(ctypes.byref(lv), ctypes.byref(self), ctypes.sizeof(HashLevels))
            lv.bwd_plan = bits
            cache[bits] = lv
        return lv


BWD_PLAN_DETERMINISTIC, BWD_PLAN_CONCENTRATED = 1, 2


def _render_fields():
    P, I, F, LL = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_longlong
    ptrs = lambda *names: [(k, P) for k in names]
    return (ptrs("rays_o", "rays_d", "hits_t", "noise") + [("n_rays", I), ("max_samples", I)]
            + ptrs("bitfield", "coarse") + [("rebuild_coarse", I), ("cascades", I), ("grid_size", I)]
            + [("scale", F), ("exp_step_factor", F), ("T_threshold", F)]
            + [("table", P), ("table_kind", I), ("enc_pairs", I), ("levels", P), ("lo", F), ("hi", F)]
            + [("w", P * 5), ("wpack", P), ("cap", LL)]
            + ptrs("stage", "march_ctr", "xyzs", "dirs", "deltas", "ts", "enc", "sigmas", "rgbs", "ws")
            + ptrs("rays_a", "total", "vr_per_ray", "opacity", "depth", "rgb")
            + ptrs("g_opacity", "g_depth", "g_rgb", "g_ws")
            + ptrs("d_sigmas", "d_rgbs", "d_enc", "live_off", "live_idx", "live_total")
            + [("workspace", P), ("workspace_bytes", LL), ("force_atomic", I), ("reserved", I)]
            + ptrs("dW_parts", "live_zero")
            + [("dW", P), ("dtable", P), ("dtable_bytes", LL)]
            + [("bg", F), ("clear_grads", I), ("rgb_out", P)])


class RenderArgs(ctypes.Structure):
    """Mirror of `ngp_render_args` (include/ngp_hip.h): the argument block of ngp_render_train_fwd / _bwd."""
    _fields_ = _render_fields()


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _flags():
    return HIPCC_FLAGS + os.environ.get("NGP_HIPCC_EXTRA", "").split()   # e.g. -DNGP_BWD_DIAG for the timing experiments in profiles/microbench


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _sources() + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "ngp_hip.h"), os.path.join(INCLUDE, "ngp_hip_experimental.h")]
    if any(os.path.getmtime(d) > t for d in deps if os.path.exists(d)):
        return True
    # a library built with other flags (a -D... diagnostic build left in the tree) is stale too; no stamp = a prebuilt library that
    # travelled without one: trusted when no extra flags are asked for
    if os.path.exists(FLAGS_STAMP):
        return open(FLAGS_STAMP).read().split() != _flags()
    return bool(os.environ.get("NGP_HIPCC_EXTRA", "").split())


def build(force=False, verbose=False):
    """Cross-compile every HIP source for gfx950 into the in-tree libngp_hip.so (works without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libngp_hip.so")
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc] + _flags() + ["-o", tmp] + _sources()
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    with open(FLAGS_STAMP, "w") as f:
        f.write(" ".join(_flags()))
    return LIB_PATH


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_LV = ctypes.POINTER(HashLevels)

# name -> argtypes (restype is always int); must list every symbol include/ngp_hip.h and include/ngp_hip_experimental.h declare
SIGNATURES = {
    "ngp_abi_version": [],
    "ngp_hash_levels_init": [_LV, ctypes.c_double, _I, ctypes.c_double, ctypes.c_double, _I],
    "ngp_ray_aabb": [_P, _P, _F, _I, _P, _P],
    "ngp_march_train_count": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P],
    "ngp_bitfield_coarsen": [_P, _I, _I, _P, _P],
    "ngp_march_train_count_ex": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P],
    "ngp_march_train_scan": [_P, _I, _P, _P, _P],
    "ngp_march_train_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ngp_march_train_fused_cap": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ngp_march_train_fused_rng": [_P, _P, _P, _P, _P, ctypes.c_ulonglong, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ngp_march_train_fused_shaped": [_P, _P, _P, _P, _P, _P, ctypes.c_ulonglong, _I, _I, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P],
    "ngp_rng_uniform": [ctypes.c_ulonglong, _I, _P, _P],
    "ngp_hash_fwd_list": [_P, _P, _I, _LV, _I, _P, _P, _I, _F, _F, _I, _P, _P],
    "ngp_mlp_fwd_list": [_P, _P, _P, _I, _P, _P, _I, _P, _P, _P],
    "ngp_live_list": [_P, _P, _I, _P, _P, _P, _P],
    "ngp_mse_loss_grad_rays": [_P, _P, _P, _F, _I, _P, _P, _P, _P, _P],
    "ngp_chunk_schedule": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "ngp_march_train_write": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "ngp_march_test": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P],
    "ngp_hash_fwd_f32": [_P, _P, _LV, _I, _P, _P],
    "ngp_hash_bwd_f32": [_P, _P, _LV, _I, _P, _P],
    "ngp_hash_fwd_f32_ex": [_P, _P, _LV, _I, _P, _I, _F, _F, _I, _P, _P],
    "ngp_hash_fwd_bf16_ex": [_P, _P, _LV, _I, _P, _I, _F, _F, _I, _P, _P],
    "ngp_hash_bwd_f32_ex": [_P, _P, _LV, _I, _P, _I, _F, _F, _I, _P, _P, _P],
    "ngp_hash_fwd_f16": [_P, _P, _LV, _I, _P, _P],
    "ngp_hash_bwd_f16": [_P, _P, _LV, _I, _P, _P],
    "ngp_hash_fwd_f16_ex": [_P, _P, _LV, _I, _P, _I, _F, _F, _I, _P, _P],
    "ngp_hash_bwd_f16_ex": [_P, _P, _LV, _I, _P, _I, _F, _F, _I, _P, _P, _P],
    "ngp_check_finite_f16": [_P, ctypes.c_longlong, _P, _P],
    "ngp_live_compact": [_P, _P, _I, _P, _P, _P, _P],
    "ngp_mlp_bwd_live": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P],
    "ngp_mlp_dw_parts_max": [],
    "ngp_mlp_bwd_live_parts": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P],
    "ngp_mlp_dw_reduce": [_P, _I, _P, _P],
    "ngp_hash_bwd_f32_live": [_P, _P, _LV, _I, _P, _P, _I, _F, _F, _I, _P, _P, _P],
    "ngp_hash_bwd_f16_live": [_P, _P, _LV, _I, _P, _P, _I, _F, _F, _I, _P, _P, _P],
    "ngp_hash_bwd_sliced_workspace": [_LV, _I],
    "ngp_hash_bwd_sliced_debug": [_P],
    "ngp_hash_bwd_sliced_plan": [_LV, _P, _I, _P, _P, _P, _P, _P],
    "ngp_hash_bwd_sliced_prep": [_P, _LV, _I, _P, _P, _I, _F, _F, _P, ctypes.c_longlong, _P],
    "ngp_hash_bwd_sliced_main": [_P, _LV, _I, _P, _I, _P, _P, _P, ctypes.c_longlong, _P],
    "ngp_hash_bwd_sliced_main_slabs": [_P, _LV, _I, _P, _I, _P, _I, _P, _P, ctypes.c_longlong, _P, _I, _P, _P],
    "ngp_hash_bwd_sliced_main_levels": [_P, _LV, _I, _P, _I, _P, _P, _P, ctypes.c_longlong, ctypes.c_uint32, _I, _P],
    "ngp_hash_bwd_sliced_adam_prefix": [_LV],
    "ngp_hash_bwd_sliced_main_adam_step": [_P, _LV, _I, _P, _I, _P, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _F,
                                           _F, _F, _F, _I, _P],
    "ngp_hash_bwd_sliced_main_adam": [_P, _LV, _I, _P, _I, _P, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P],
    "ngp_hash_bwd_sliced_main_f16": [_P, _LV, _I, _P, _I, _P, _P, _P, ctypes.c_longlong, _P],
    "ngp_hash_bwd_f32_sliced": [_P, _P, _LV, _I, _P, _P, _I, _F, _F, _I, _P, _P, _P, ctypes.c_longlong, _P],
    "ngp_sh16_fwd": [_P, _I, _P, _P],
    "ngp_sh16_bwd": [_P, _P, _I, _P, _P],
    "ngp_composite_train_fwd": [_P, _P, _I, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P],
    "ngp_composite_train_bwd": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P, _P],
    "ngp_composite_train_fwd_bg": [_P, _P, _I, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P, _F, _P],
    "ngp_composite_train_bwd_bg": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P, _F, _P],
    "ngp_composite_train_fused": [_P, _P, _I, _P, _P, _P, _P, _F, _P, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ngp_composite_train_fused_live": [_P, _P, _I, _P, _P, _P, _P, _F, _P, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ngp_composite_test": [_P, _P, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P],
    "ngp_mlp_wpack_halfs": [],
    "ngp_mlp_pack": [_P, _P, _P, _P, _P, _I, _P, _P],
    "ngp_mlp_fwd": [_P, _P, _P, _I, _P, _P, _P],
    "ngp_mlp_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _P],
    "ngp_mlp_fwd_ex": [_P, _P, _P, _I, _P, _I, _P, _P, _P],
    "ngp_mlp_bwd_ex": [_P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P],
    "ngp_mse_loss_grad": [_P, _P, _P, _F, _I, _P, _P, _P, _P],
    "ngp_event_create": [ctypes.POINTER(ctypes.c_void_p)],
    "ngp_event_record": [_P, _P],
    "ngp_stream_wait_event": [_P, _P],
    "ngp_event_destroy": [_P],
    "ngp_event_synchronize": [_P],
    "ngp_stream_create_low_priority": [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "ngp_stream_destroy": [_P],
    "ngp_render_train_fwd": [_P, _P],
    "ngp_render_train_bwd": [_P, _P],
    "ngp_host_alloc": [_P, ctypes.c_longlong],
    "ngp_host_free": [_P],
    "ngp_copy_to_host_async": [_P, _P, ctypes.c_longlong, _P],
    "ngp_train_prologue": [_P, _P, _F, _F, _I, _F, _F, _F, _F, _I, _P],
    "ngp_train_prologue_reduce": [_P, _P, _F, _F, _I, _F, _F, _F, _F, _I, _P, _I, _P, _P],
    "ngp_adam_amp_prologue": [_P, _P, _P, _P, _F, _F, _F, _P],
    "ngp_adam_step": [_P, _P, _P, _P, ctypes.c_longlong, _P, _P, _F, _F, _F, _P],
    "ngp_adam_multi": [_I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P],
    "ngp_check_finite_multi": [_I, _P, _P, _P, _P],
    "ngp_adam_amp_check_prologue": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P],
    "ngp_adam_step_bf16": [_P, _P, _P, _P, ctypes.c_longlong, _P, _P, _F, _F, _F, _P, _P],
    "ngp_cast_f32_bf16": [_P, _P, ctypes.c_longlong, _P],
    "ngp_adam_all": [_P, _P, _P, _P, ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P],
    "ngp_adam_all_ex": [_P, _P, _I, _P, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P],
    "ngp_adam_mlp_pack": [_P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P],
    "ngp_distortion_fwd": [_P, _P, _P, _P, _I, _P, _P, _P, _P],
    "ngp_distortion_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "ngp_get_rays": [_P, _P, _I, _I, _P, _P, _P],
    "ngp_stage_batch": [_P, _P, _P, _P, _P, _P, _I, _P],
    "ngp_sample_rays": [_P, _P, _P, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P],
    "ngp_occ_compact": [_P, _F, _I, _P, _P, _P, _P],
    "ngp_sorted_uniforms": [_P, _I, _I, _P, _P, _P],
    "ngp_occ_sample": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P],
    "ngp_occ_all_cells": [_P, _I, _I, _F, _F, _P, _P],
    "ngp_occ_scatter": [_P, _P, _I, _P, _P],
    "ngp_occ_scatter_max": [_P, _P, _I, _P, _P],
    "ngp_occ_stats_floats": [],
    "ngp_occ_merge": [_P, _P, _F, _I, _P, _P],
    "ngp_occ_pack": [_P, _P, _F, _I, _P, _P],
    "ngp_p2p_max_peers": [],
    "ngp_p2p_push": [_P, ctypes.c_longlong, _I, _I, _P, _P, ctypes.c_longlong, _I, _I, _I, _P, _P],
    "ngp_p2p_wait": [_P, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _F, _I, _P, _P, _P],
    "ngp_morton3d": [_P, _I, _P, _P],
    "ngp_morton3d_invert": [_P, _I, _P, _P],
    "ngp_packbits": [_P, _F, _I, _P, _P],
}

# declared in include/ngp_hip_experimental.h: exported and typed like the rest, called by no default path of the package
EXPERIMENTAL = {"ngp_adam_all", "ngp_adam_step", "ngp_adam_step_bf16", "ngp_composite_train_fused", "ngp_hash_bwd_f16_ex", "ngp_hash_bwd_f32_ex",
                "ngp_hash_bwd_sliced_debug", "ngp_hash_bwd_sliced_plan", "ngp_march_train_count", "ngp_mlp_bwd_ex",
                "ngp_hash_bwd_sliced_main_adam", "ngp_hash_bwd_sliced_main_levels", "ngp_p2p_max_peers", "ngp_p2p_push", "ngp_p2p_wait"}
_LONGLONG_RESULT = {"ngp_hash_bwd_sliced_workspace", "ngp_hash_bwd_sliced_adam_prefix"}       # byte counts; every other entry point returns an int status
_lib = None


def load():
    """Load libngp_hip.so (after torch, so both share torch's libamdhip64.so.7) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must be imported first: the HIP runtime SONAME is then already resolved)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libngp_hip.so is not built (%s). Run `python __graft_entry__.py` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name in _LONGLONG_RESULT else ctypes.c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d (negated hipError_t, or -1 for bad arguments)" % (what, rc))
