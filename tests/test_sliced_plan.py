"""Host logic of the LDS-sliced scatter-add (csrc/hash_bwd_lds.hip: build_plan), no GPU: the task plan covers every
(level, slice, sample-range replica) exactly once, the per-XCD queues tile the task array, the level classes (one-slice levels,
run pre-summing, replication) follow the level table -- for the C2 table, the C3 one (max_res 4096), a small table and tables the
formulation cannot express."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from ngp_hip import lib, ops  # noqa: E402

SLICE = 8192


def plan_of(lv):
    L = lib.load()
    tasks = np.zeros(1536, np.uint16)
    xoff, xlen = np.zeros(8, np.uint16), np.zeros(8, np.uint16)
    nrep = np.zeros(16, np.uint8)
    mm, sm = ctypes.c_uint32(0), ctypes.c_uint32(0)
    n = L.ngp_hash_bwd_sliced_plan(ctypes.byref(lv), tasks.ctypes.data_as(ctypes.c_void_p), tasks.size,
                                   xoff.ctypes.data_as(ctypes.c_void_p), xlen.ctypes.data_as(ctypes.c_void_p),
                                   nrep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mm), ctypes.byref(sm))
    return n, tasks, xoff, xlen, nrep, mm.value, sm.value


@pytest.mark.parametrize("log2_t,max_res", [(19, 1024), (19, 4096), (14, 512), (21, 2048)])
def test_plan_covers_every_slice_replica_once(log2_t, max_res):
    lv = ops.make_levels(2**log2_t, 16, 16, max_res, 2)
    n, tasks, xoff, xlen, nrep, merge_mask, single_mask = plan_of(lv)
    sizes = [int(lv.map_size[l]) for l in range(16)]
    if max(sizes) > 64 * SLICE:                                  # a level of more than 64 slices: not expressible
        assert n == -2
        return
    ns = [(s + SLICE - 1) // SLICE for s in sizes]
    assert n == sum(a * int(b) for a, b in zip(ns, nrep)) and 0 < n <= 1536
    seen = set()
    for t in tasks[:n]:
        level, sl, rep = int(t) & 0xf, (int(t) >> 4) & 0x3f, (int(t) >> 10) & 0x3f
        assert sl < ns[level] and rep < nrep[level]
        assert (level, sl, rep) not in seen
        seen.add((level, sl, rep))
    assert len(seen) == n
    # the eight XCD queues tile the task array
    assert int(xoff[0]) == 0 and int(xlen.sum()) == n
    for x in range(7):
        assert int(xoff[x + 1]) == int(xoff[x]) + int(xlen[x])
    # no XCD is left without work, none holds more than twice its share
    assert int(xlen.min()) > 0 and int(xlen.max()) <= 2 * ((n + 7) // 8) + 32
    bfhl = int(lv.begin_fast_hash_level)
    for l in range(16):
        assert bool((single_mask >> l) & 1) == (ns[l] == 1)
        if (merge_mask >> l) & 1:
            assert l < bfhl and int(lv.resolution[l]) <= 128     # run pre-summing: dense coarse levels only
        if l >= bfhl and ns[l] == 64:
            assert nrep[l] == 1                                   # a full hashed level: one owner per slice
        if l < bfhl and ns[l] > 1:
            assert nrep[l] >= 4                                   # dense slices follow the scene: never fewer than 4 sample ranges
    # a level's owners sit together inside an XCD queue (they share position / gradient lines in that XCD's L2)
    for x in range(8):
        q = [int(t) & 0xf for t in tasks[int(xoff[x]):int(xoff[x]) + int(xlen[x])]]
        runs = sum(1 for i in range(1, len(q)) if q[i] != q[i - 1]) + 1
        assert runs <= len(set(q)) + 2, (x, q)


def test_plan_refuses_other_feature_counts():
    lv = ops.make_levels(2**19, 16, 16, 1024, 4)
    assert plan_of(lv)[0] == -2


def test_flush_adam_prefix_and_deterministic_plan():
    """Round 5: the levels whose slices have one owner (nrep == 1) are a suffix of the level table; ngp_hash_bwd_sliced_adam_prefix
    names where it starts (in table floats).  In deterministic mode every level has one owner per slice: prefix 0, tasks = slices."""
    L = lib.load()
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)
    try:
        n, tasks, xoff, xlen, nrep, mm, sm = plan_of(lv)
        first = min(l for l in range(16) if nrep[l] == 1)
        assert all(nrep[l] == 1 for l in range(first, 16)) and all(nrep[l] > 1 for l in range(first))
        assert L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lv)) == 2 * int(lv.offset[first])
        lv_d = lv.with_plan(lib.BWD_PLAN_DETERMINISTIC)
        n_d, tasks_d, _, xlen_d, nrep_d, mm_d, _ = plan_of(lv_d)
        sizes = [int(lv.map_size[l]) for l in range(16)]
        assert all(int(r) == 1 for r in nrep_d) and n_d == sum((s + SLICE - 1) // SLICE for s in sizes) == int(xlen_d.sum())
        assert mm_d == mm                                           # run pre-summing stays (its grouping is what becomes order-free)
        assert L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lv_d)) == 0
    finally:
        pass
    assert plan_of(lv)[0] == n
    small = ops.make_levels(2**15, 16, 16, 512, 2)                  # no level of 64 slices: nothing for the flush to own
    assert L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(small)) == -2


def test_concentrated_scene_plan():
    """Round 5 (round 6: the mode is the NGP_BWD_PLAN_CONCENTRATED bit of the level table) -- hashed levels up to resolution 256 get three sample-range replicas per slice (a
    scene that fills a small part of its box loads a few of their owners with several times the mean), the plan still fits the
    kernel argument, the flush-Adam set shrinks to the levels that keep one owner per slice, and switching the mode off restores the
    default plan.  On the C2 table (levels up to resolution 256 are dense there or already replicated) the hashed levels it touches are
    the first hashed ones only."""
    L = lib.load()
    c3 = ops.make_levels(2**19, 16, 16, 4096, 2)
    bfhl = int(c3.begin_fast_hash_level)
    n0, _, _, _, nrep0, mm0, _ = plan_of(c3)
    pre0 = L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(c3))
    c3c = c3.with_plan(lib.BWD_PLAN_CONCENTRATED)
    try:
        n1, tasks, xoff, xlen, nrep1, mm1, _ = plan_of(c3c)
        assert 0 < n1 <= 1536 and int(xlen.sum()) == n1
        hot = [l for l in range(bfhl, 16) if int(c3.resolution[l]) <= 256]
        assert hot and all(int(nrep1[l]) == 3 for l in hot)
        assert all(int(nrep1[l]) == 1 for l in range(bfhl, 16) if l not in hot)
        assert all(int(nrep1[l]) >= 4 for l in range(bfhl) if int(c3.map_size[l]) > SLICE)       # dense levels keep >= 4 sample ranges
        assert mm1 == mm0                                                                       # run pre-summing: dense levels only, as before
        first = max(hot) + 1
        assert L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(c3c)) == 2 * int(c3.offset[first]) > pre0
        # every (level, slice, replica) exactly once
        seen = set(int(t) for t in tasks[:n1])
        assert len(seen) == n1
    finally:
        pass
    n2, _, _, _, nrep2, mm2, _ = plan_of(c3)
    assert n2 == n0 and list(nrep2) == list(nrep0) and mm2 == mm0
    assert L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(c3)) == pre0


def test_plan_sweep_over_level_tables_and_modes():
    """Every level table the encoder can be built with, in every plan mode (default, deterministic, concentrated, both): the builder
    either refuses (-2) or returns a plan of at most 1536 tasks (the kernel-argument bound) that names every (level, slice, replica)
    once, with one owner per slice in deterministic mode, and a flush-Adam prefix that is exactly the start of the one-owner suffix."""
    import itertools
    L = lib.load()
    n_plans = 0
    try:
        for det, conc in itertools.product((0, 1), (0, 1)):
            bits = det * lib.BWD_PLAN_DETERMINISTIC + conc * lib.BWD_PLAN_CONCENTRATED
            for log2_t, nl, base, max_res in itertools.product((12, 15, 17, 19, 20, 22), (1, 4, 8, 16), (4, 16, 64), (64, 512, 2048, 4096, 16384)):
                lv = ops.make_levels(2**log2_t, nl, base, max_res, 2).with_plan(bits)
                n, tasks, xoff, xlen, nrep, mm, sm = plan_of(lv)
                if n < 0:
                    assert n == -2
                    continue
                n_plans += 1
                sizes = [int(lv.map_size[l]) for l in range(nl)]
                assert 0 < n <= 1536 and int(xlen.sum()) == n and len(set(int(t) for t in tasks[:n])) == n
                assert n == sum(((s + SLICE - 1) // SLICE) * int(nrep[l]) for l, s in enumerate(sizes))
                if det:
                    assert all(int(nrep[l]) == 1 for l in range(nl))
                pre = L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lv))
                if pre >= 0:
                    first = min(l for l in range(nl) if all(int(nrep[k]) == 1 for k in range(l, nl)))
                    assert pre == 2 * int(lv.offset[first])
                else:
                    assert pre == -2
    finally:
        pass
    assert n_plans > 500
