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
    assert [v[0] for v in b.COMM_VARIANTS] == ["bf16-comm+bf16-table", "no-shard-all-reduce", "overlap-8,0"]
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


def test_scatter_mode_sets_the_library_switches_every_time():
    """FusedTrainer._scatter_mode: the key under which the trainer caches its flush-Adam prefix names the plan mode the library is
    REALLY in afterwards -- it sets both per-thread switches on every call instead of trusting a Python-side copy of them (another
    caller on the thread may have flipped them in between).  Host logic only: the C switches are plain thread-local flags."""
    from types import SimpleNamespace
    from ngp_hip import lib
    from ngp_hip.trainer import FusedTrainer
    L = lib.load()
    try:
        for det, conc in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0)):
            stub = SimpleNamespace(L=L, deterministic=bool(det), _concentrated=bool(conc))
            # somebody else leaves the opposite setting behind
            L.ngp_hash_bwd_sliced_deterministic(1 - det)
            L.ngp_hash_bwd_sliced_concentrated(1 - conc)
            assert FusedTrainer._scatter_mode(stub) == det + 2 * conc
            assert L.ngp_hash_bwd_sliced_deterministic(det) == det and L.ngp_hash_bwd_sliced_concentrated(conc) == conc
    finally:
        L.ngp_hash_bwd_sliced_deterministic(0)
        L.ngp_hash_bwd_sliced_concentrated(0)
