"""Host-side logic added in round 5 that needs no GPU: the chunked forward's round table, the exchange-variant record of bench.py."""
import importlib.util
import os

from conftest import ROOT


def test_chunk_rounds_cover_max_samples_on_the_64_grid():
    from ngp_hip.trainer import FusedTrainer
    assert FusedTrainer.chunk_rounds(1024) == [(0, 64, 0), (64, 64, 0), (128, 128, 64), (256, 256, 128), (512, 512, 256)]
    assert FusedTrainer.chunk_rounds(128) == [(0, 64, 0), (64, 64, 0)]
    for ms in (128, 192, 256, 640, 1024, 2048):
        r = FusedTrainer.chunk_rounds(ms)
        assert r[0][0] == 0 and sum(l for _, l, _ in r) == ms
        for k, (b, l, pb) in enumerate(r):
            assert b % 64 == 0 and l % 64 == 0 and l > 0
            assert pb == (r[k - 1][0] if k else 0) and (k == 0 or b == r[k - 1][0] + r[k - 1][1])


def _bench():
    spec = importlib.util.spec_from_file_location("bench_r05", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_comm_record_bus_bandwidth_and_model():
    """bench.py --gpus N attaches one record per exchange variant: bus bytes in RCCL's convention ((N-1)/N of a reduce-scatter /
    all-gather payload, twice that of an all-reduce), the bandwidth comm_ms implies, and DESIGN section 7's model on the measured
    communication-free step."""
    b = _bench()
    assert [v[0] for v in b.COMM_VARIANTS] == ["bf16-comm+bf16-table", "no-shard-all-reduce", "overlap-8,0", "p2p-direct"]
    o = {"value": 1e8, "ms_per_step": 0.8, "steps": 20, "comm_ms": 0.3, "exposed_comm_ms": 0.25, "ms_per_step_comm_stubbed": 0.55,
         "comm_breakdown_ms": {"reduce_scatter_table_grad": 0.14, "all_gather_table": 0.15, "all_reduce_mlp_grad_and_flag": 0.01},
         "comm_bytes_per_rank_per_step": {"reduce_scatter_table_grad": 45_680_256, "all_gather_table": 45_680_256,
                                          "all_reduce_mlp_grad_and_flag": 37_648},
         "config": {"global_batch": 65536, "parallelism": "x"}, "live_samples_per_step": 1.0}
    r = b._comm_record("n", "w", [], {}, o, 8)
    bus_bytes = (45_680_256 * 2 + 2 * 37_648) * 7 / 8
    assert abs(r["bus_bandwidth_GBs"] - bus_bytes / 0.3e-3 / 1e9) < 1e-6
    m = r["design7_model_at_bus_bandwidth"]["300_GBs"]
    assert abs(m["ms_per_step"] - (0.55 + bus_bytes / 300e9 * 1e3)) < 1e-9
    assert abs(m["rays_per_s"] - 65536 / (m["ms_per_step"] * 1e-3)) < 1e-3
    o2 = dict(o, comm_bytes_per_rank_per_step={"all_reduce_flat_bucket": 45_717_904}, comm_breakdown_ms={"all_reduce_flat_bucket": 0.3})
    r2 = b._comm_record("n", "w", ["--no-shard"], {}, o2, 2)
    assert abs(r2["bus_bandwidth_GBs"] - 2 * 45_717_904 * 0.5 / 0.3e-3 / 1e9) < 1e-6


def test_scatter_mode_travels_in_the_level_table():
    """Round 6: the scatter-add's plan mode is no longer per-thread state of the library.  FusedTrainer._scatter_mode
    returns NGP_BWD_PLAN_* bits, HashLevels.with_plan puts them into a (cached) copy of the level table, and the library's plan follows
    the table it is handed -- two tables with different bits give their own plans in any call order.  Host logic only."""
    import ctypes
    from types import SimpleNamespace
    from ngp_hip import lib, ops
    from ngp_hip.trainer import FusedTrainer
    L = lib.load()
    assert not hasattr(L, "ngp_hash_bwd_sliced_deterministic") and not hasattr(L, "ngp_hash_bwd_sliced_concentrated")
    lv = ops.make_levels(2**19, 16, 16, 4096, 2)
    assert lv.bwd_plan == 0 and lv.with_plan(0) is lv

    def nrep_of(t):
        nrep = (ctypes.c_uint8 * 16)()
        assert L.ngp_hash_bwd_sliced_plan(ctypes.byref(t), None, 0, None, None, nrep, None, None) > 0
        return list(nrep)
    base = nrep_of(lv)
    seen = {}
    for det, conc in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (0, 1)):
        stub = SimpleNamespace(deterministic=bool(det), _concentrated=bool(conc))
        bits = FusedTrainer._scatter_mode(stub)
        assert bits == det * lib.BWD_PLAN_DETERMINISTIC + conc * lib.BWD_PLAN_CONCENTRATED
        t = lv.with_plan(bits)
        assert t.bwd_plan == bits and lv.with_plan(bits) is t and lv.bwd_plan == 0           # cached copy, the original untouched
        assert bytes(t)[:ctypes.sizeof(t) - 4] == bytes(lv)[:ctypes.sizeof(lv) - 4]
        got = nrep_of(t)
        assert seen.setdefault(bits, got) == got                                              # a function of the table, not of history
        if det:
            assert all(r == 1 for r in got)
        assert nrep_of(lv) == base                                                            # the default table still gets the default plan
    assert seen[lib.BWD_PLAN_CONCENTRATED] != base


def test_coarse_table_validity_key_is_shared_and_never_aliases():
    """Round 6: the 8^3-block occupancy shortcut table of an arena has ONE validity key, kept on the arena, that both
    FusedTrainer and the fused render() consult -- and the key names the render configuration by a serial that is never reused, so a
    new model whose bitfield lands on a freed bitfield's address with an equal version count still rebuilds (the first cut of this
    change keyed on the address alone and marched a fresh model against the previous model's table)."""
    import torch
    from types import SimpleNamespace
    from ngp_hip.fused import RenderConfig, TrainArena
    A = TrainArena(torch.device("cpu"), 4, 4)
    bits = torch.zeros(128**3 // 8, dtype=torch.uint8)

    def cfg_for(serial):
        return SimpleNamespace(serial=serial, bitfield=bits, cascades=1, grid_size=128)
    c1 = cfg_for(1)
    buf, stale = A.coarse_state(c1)
    assert stale and buf.numel() == 128**3 // 512 // 32
    assert not A.coarse_state(c1)[1]                                                # second consumer of the same configuration: valid
    bits.add_(1)                                                                    # an in-place write moves the version counter
    assert A.coarse_state(c1)[1] and not A.coarse_state(c1)[1]
    c2 = cfg_for(2)                                                                 # another model, same address, same version
    assert A.coarse_state(c2)[1] and not A.coarse_state(c2)[1]
    assert A.coarse_state(c1)[1]                                                    # ... and back: the buffer now holds c2's table
    TrainArena._cache[("test", 4, 4)] = A
    try:
        TrainArena.invalidate_coarse()
        assert A.coarse_state(c1)[1]
    finally:
        del TrainArena._cache[("test", 4, 4)]
    # RenderConfig hands out fresh serials
    m = SimpleNamespace(scale=0.5, cascades=1, grid_size=128, density_bitfield=bits, half_opt=False,
                        pos_encoder=SimpleNamespace(levels_struct=__import__("ngp_hip.ops", fromlist=["x"]).make_levels(2**19, 16, 16, 1024, 2)))
    a, b = RenderConfig(m, 0.0, 1e-4, 1024), RenderConfig(m, 0.0, 1e-4, 1024)
    assert a.serial != b.serial and a.levels.bwd_plan == 0
    m3 = SimpleNamespace(**{**m.__dict__, "scale": 16.0, "cascades": 6})
    from ngp_hip import lib
    assert RenderConfig(m3, 1.0 / 256, 1e-4, 1024).levels.bwd_plan == lib.BWD_PLAN_CONCENTRATED     # multi-cascade scene: concentrated plan on the drop-in path too
