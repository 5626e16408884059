"""Device-resident occupancy-grid update (reference modules/networks.py:181-209,255-290), no host round trips.

Same algorithm as the reference (and as NGP.update_density_grid's torch formulation, which stays as the generic path):
per cascade, either all cells (warm-up) or M = G^3/4 uniform cells + M cells drawn from the occupied set; a jittered point
per cell; density there; decay/max merge; bitfield threshold = min(mean positive density, density_threshold)."""
import ctypes
import os

import torch

from . import lib as _lib_mod
from .lib import check
from .ops import _ptr, _stream, _touched


class OccupancyUpdater:

    def __init__(self, model):
        self.model = model
        self.L = _lib_mod.load()
        G3 = model.grid_size**3
        dev = model.density_grid.device
        self.dev = dev
        self.M = G3 // 4
        n_max = max(G3, 2 * self.M)
        f32 = dict(device=dev, dtype=torch.float32)
        self.list = torch.empty(G3, device=dev, dtype=torch.int32)
        self.count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.scratch = torch.empty(1024, device=dev, dtype=torch.int32)
        self.indices = torch.empty(2 * self.M, device=dev, dtype=torch.int32)
        self.xyzs = torch.empty(n_max, 3, **f32)
        self.enc = torch.empty(n_max, 32, **f32)
        self.sigmas = torch.empty(n_max, **f32)
        self.tmp = torch.empty(model.cascades, G3, **f32)
        self.stats = torch.zeros(self.L.ngp_occ_stats_floats(), **f32)        # [0] sum, [1] count, then the merge kernel's per-block partials
        self.wpack = torch.empty(self.L.ngp_mlp_wpack_halfs(), device=dev, dtype=torch.float16)
        lvs = getattr(getattr(model, "pos_encoder", None), "levels_struct", None)     # (None: a grid driven with density_fn only)
        self.enc_pairs = 1 if (lvs is not None and lvs.n_levels == 16 and lvs.n_features == 2) else 0
        self._su_work = self._su_out = None

    @torch.no_grad()
    def update(self, density_threshold, warmup=False, decay=0.95, jitter=None, uniforms=None, density_fn=None):
        """One occupancy-grid update.  The three optional hooks exist for the fixture tests (tests/test_gpu_golden.py holds this
        class to vectors produced by the reference's own NGP.update_density_grid): `jitter(c, n)` -> [n, 3] uniforms replacing
        torch.rand for cascade c's in-cell jitter; `uniforms(c)` -> (u_cell [M], u_pick [M]) replacing the sorted uniforms that
        choose the cells; `density_fn(c, xyzs [n, 3], indices [n] or None)` -> sigmas [n] replacing the hash-grid + MLP density."""
        m, L, st = self.model, self.L, _stream()
        det = bool(getattr(m, "_ngp_deterministic", False)) or os.environ.get("NGP_DETERMINISTIC", "0") == "1"
        G, G3, C = m.grid_size, m.grid_size**3, m.cascades
        grid = m.density_grid
        if not grid.is_contiguous():
            raise ValueError("density_grid must be contiguous")
        if density_fn is None:
            lv = m.pos_encoder.levels_struct
            ws = m._mlp_weights()
            check(L.ngp_mlp_pack(*[_ptr(w) for w in ws], self.enc_pairs, _ptr(self.wpack), st), "ngp_mlp_pack")
        self.tmp.zero_()                                                       # density_grid_tmp = zeros_like, networks.py:261
        lo, hi = -float(m.scale), float(m.scale)
        for c in range(C):
            s = min(2.0**(c - 1), float(m.scale))
            hg = s / G
            grid_c = grid[c]
            tmp_c = self.tmp[c]
            if warmup:
                n = G3
                u_jit = torch.rand(n, 3, device=self.dev) if jitter is None else jitter(c, n).contiguous().float()
                check(L.ngp_occ_all_cells(_ptr(u_jit), n, G, s, hg, _ptr(self.xyzs), st), "ngp_occ_all_cells")
                idx_ptr = _ptr(None)
            else:
                n = 2 * self.M
                # order statistics of M iid uniforms without sorting: normalised partial sums of M+1 unit exponentials
                # (exact in distribution).  Ascending uniforms -> ascending Morton codes / list positions -> neighbouring
                # encoder queries share hash-grid lines (the 1 M-point encode is gather-bound: 900 -> ~540 us).
                # (ngp_sorted_uniforms: row scans of 1024 + row offsets, two launches; the same formula took ten torch kernels)
                rows = (self.M + 1 + 1023) // 1024
                u_raw = torch.rand(2 * rows * 1024, device=self.dev)
                if self._su_work is None:
                    self._su_work = torch.empty(2 * (rows * 1024 + rows), device=self.dev, dtype=torch.float32)
                    self._su_out = torch.empty(2, self.M, device=self.dev, dtype=torch.float32)
                check(L.ngp_sorted_uniforms(_ptr(u_raw), self.M, 2, _ptr(self._su_work), _ptr(self._su_out), st), "ngp_sorted_uniforms")
                u_cell, u_pick = self._su_out[0], self._su_out[1]
                if uniforms is not None:
                    u_cell, u_pick = [u.contiguous().float() for u in uniforms(c)]
                u_jit = torch.rand(n * 3, device=self.dev) if jitter is None else jitter(c, n).contiguous().float()
                check(L.ngp_occ_compact(_ptr(grid_c), float(density_threshold), G3, _ptr(self.list), _ptr(self.count), _ptr(self.scratch), st),
                      "ngp_occ_compact")
                check(L.ngp_occ_sample(_ptr(u_cell), _ptr(u_pick), _ptr(u_jit), _ptr(self.list), _ptr(self.count), self.M, G, s, hg,
                                       _ptr(self.indices), _ptr(self.xyzs), st), "ngp_occ_sample")
                idx_ptr = _ptr(self.indices)
            if density_fn is not None:
                self.sigmas[:n].copy_(density_fn(c, self.xyzs[:n], None if warmup else self.indices[:n]))
            elif getattr(m, "half_opt", False):                                # half2 encoder: its own arithmetic on the f16 copy
                check(L.ngp_hash_fwd_f16_ex(_ptr(self.xyzs), _ptr(m.pos_encoder.table_f16()), ctypes.byref(lv), n, _ptr(None), 1, lo, hi,
                                            self.enc_pairs, _ptr(self.enc), st), "ngp_hash_fwd_f16_ex")
            elif getattr(m.pos_encoder, "table_dtype", torch.float32) == torch.bfloat16:
                check(L.ngp_hash_fwd_bf16_ex(_ptr(self.xyzs), _ptr(m.pos_encoder.table_bf16()), ctypes.byref(lv), n, _ptr(None), 1, lo,
                                             hi, self.enc_pairs, _ptr(self.enc), st), "ngp_hash_fwd_bf16_ex")
            else:
                check(L.ngp_hash_fwd_f32_ex(_ptr(self.xyzs), _ptr(m.pos_encoder.hash_table), ctypes.byref(lv), n, _ptr(None), 1, lo, hi,
                                            self.enc_pairs, _ptr(self.enc), st), "ngp_hash_fwd_f32_ex")
            if density_fn is None:
                check(L.ngp_mlp_fwd_ex(_ptr(self.enc), _ptr(None), _ptr(self.wpack), n, _ptr(None), self.enc_pairs, _ptr(self.sigmas),
                                       _ptr(None), st), "ngp_mlp_fwd_ex")
            # (deterministic mode, FusedTrainer.set_deterministic / NGP_DETERMINISTIC=1: of the densities drawn for one cell the largest
            # is kept instead of whichever write lands last)
            if det and not warmup:
                check(L.ngp_occ_scatter_max(idx_ptr, _ptr(self.sigmas), n, _ptr(tmp_c), st), "ngp_occ_scatter_max")
            else:
                check(L.ngp_occ_scatter(idx_ptr, _ptr(self.sigmas), n, _ptr(tmp_c), st), "ngp_occ_scatter")
        check(L.ngp_occ_merge(_ptr(grid), _ptr(self.tmp), float(decay), C * G3, _ptr(self.stats), st), "ngp_occ_merge")
        check(L.ngp_occ_pack(_ptr(grid), _ptr(self.stats), float(density_threshold), C * G3 // 8, _ptr(m.density_bitfield), st),
              "ngp_occ_pack")
        _touched(grid, m.density_bitfield)       # written through raw pointers: version-keyed caches (coarse table) must see it
