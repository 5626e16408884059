"""N>1 path on CPU: world_size-2 gloo processes. Checks the ray sharding and that averaging the per-shard gradients
with GradReducer reproduces the single-process gradient of the whole batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Toy(torch.nn.Module):
    """Same structure as the hot path's parameters: one big 'table' gathered by index + a small dense head."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.table = torch.nn.Parameter(torch.rand(1 << 21, generator=g))       # > big_numel -> own collective
        self.head = torch.nn.Linear(8, 3, bias=False)
        with torch.no_grad():
            self.head.weight.copy_(torch.rand(3, 8, generator=g))

    def forward(self, idx):
        return self.head(self.table[idx])


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd"))
    from ngp_hip.dist import GradReducer, shard_rays
    torch.manual_seed(1)                                    # identical replicas and identical global batch on every rank
    model = _Toy()
    idx = torch.randint(0, 1 << 21, (n_total, 8))
    target = torch.rand(n_total, 3)
    a, b = shard_rays(n_total, rank, world)
    loss = torch.nn.functional.mse_loss(model(idx[a:b]), target[a:b])
    loss.backward()
    GradReducer(model, world).all_reduce()
    torch.save({"table": model.table.grad.to_sparse(), "head": model.head.weight.grad, "shard": (a, b)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process(tmp_path):
    n_total, world = 4096, 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    torch.manual_seed(1)
    model = _Toy()
    idx = torch.randint(0, 1 << 21, (n_total, 8))
    target = torch.rand(n_total, 3)
    torch.nn.functional.mse_loss(model(idx), target).backward()
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert r0["shard"] == (0, 2048) and r1["shard"] == (2048, 4096)
    for r in (r0, r1):                                      # every rank ends with the same, full-batch gradient
        torch.testing.assert_close(r["table"].to_dense(), model.table.grad, rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(r["head"], model.head.weight.grad, rtol=1e-5, atol=1e-8)


def test_shard_rays_covers_batch():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd"))
    from ngp_hip.dist import shard_rays
    for n, w in ((65536, 8), (8192, 3), (5, 8)):
        spans = [shard_rays(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _sharded_worker(rank, world, port, out_path):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ngp_hip.dist import all_gather_shards, reduce_scatter_avg
    n, lr = 4096, 0.1
    g = torch.Generator().manual_seed(0)
    table = torch.randn(n, generator=g)                       # replicated parameters
    grads = torch.randn(world, n, generator=g)                # rank r's local gradient = grads[r]
    sh = n // world
    shard_grad = torch.empty(sh)
    reduce_scatter_avg(shard_grad, grads[rank].clone(), rank, world)
    table[rank * sh:(rank + 1) * sh] -= lr * shard_grad       # the optimizer touches the own shard only
    all_gather_shards(table, rank, sh, world)
    want = torch.randn(n, generator=torch.Generator().manual_seed(0)) - lr * grads.mean(0)
    ok = torch.allclose(table, want, atol=1e-6)
    both = [torch.empty_like(table) for _ in range(world)]
    dist.all_gather(both, table)
    ok = ok and all(torch.equal(both[0], b) for b in both)
    dist.destroy_process_group()
    open(out_path + ".%d" % rank, "w").write("ok" if ok else "mismatch")


def test_two_rank_sharded_exchange_equals_replicated_update(tmp_path):
    """reduce-scatter(avg) -> update of the own shard -> all-gather (FusedTrainer's N > 1 exchange, ngp_hip/dist.py) leaves
    every rank with exactly the parameters a replicated update on the averaged gradient gives (2 ranks, gloo, CPU)."""
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 400
    out = str(tmp_path / "res")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out + ".0").read() == "ok" and open(out + ".1").read() == "ok"


# ---- round 4: the overlapped exchange's level groups (ngp_hip/dist.py: make_level_groups, GroupedExchange) on CPU tensors ---------
def _c2_offsets():
    """Entry offsets of the 16 levels of the BASELINE C2 table (hash_encoder.py:183-205 arithmetic, via the C library's host code)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd"))
    from ngp_hip import ops
    lv = ops.make_levels(2**19, 16, 16.0, 1024.0, 2)
    return [int(lv.offset[l]) for l in range(16)], int(lv.total_entries) * 2


def test_level_groups_partition_the_table_for_every_world_size():
    """Every group is split into `world` chunks of a multiple of 4 floats that tile it exactly once; the union over groups is the
    padded table; the masks are disjoint and cover all levels -- for 1..8 ranks and several groupings."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd"))
    from ngp_hip.dist import make_level_groups
    offsets, nt = _c2_offsets()
    for world in (1, 2, 3, 4, 5, 8):
        unit = 4 * world
        nt_pad = (nt + unit - 1) // unit * unit
        for spec in ("0", "8,0", "12,8,0", "15,3,0", "6,0"):
            per_rank = [make_level_groups(offsets, 2, nt_pad, world, r, spec) for r in range(world)]
            groups = per_rank[0]
            assert sum(g.mask for g in groups) == 0xffff and all(a.mask & b.mask == 0 for a in groups for b in groups if a is not b)
            covered = 0
            for k, g in enumerate(groups):
                assert g.c % 4 == 0 and g.c * world >= g.b - g.a and (g.c * world == g.b - g.a) == g.aligned
                spans = [(pr[k].lo, pr[k].hi) for pr in per_rank]
                assert spans[0][0] == g.a and max(h for _, h in spans) == g.b
                for (l0, h0), (l1, h1) in zip(spans, spans[1:]):
                    assert h0 == min(l1, g.b) and (h0 - l0) % 4 == 0           # contiguous, whole float4 groups (Adam's unit)
                covered += g.b - g.a
            assert covered == nt_pad and groups[0].b == nt_pad and groups[-1].a == 0
    import pytest
    for bad in ("8", "0,8", "8,8,0", "16,0", ""):
        with pytest.raises(ValueError):
            make_level_groups(offsets, 2, nt, 2, 0, bad)


def _grouped_worker(rank, world, port, spec, comm16, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taichi-nerfs_amd"))
    from ngp_hip.dist import GroupedExchange, make_level_groups
    # a small table with the REAL structure: level sizes that are multiples of 16 floats only (so 4 * world rarely divides a group)
    sizes = [4096, 9264, 19688, 46656, 32768, 32768, 32768, 32768]
    offsets = [sum(sizes[:l]) for l in range(len(sizes))]
    nt = 2 * sum(sizes)
    unit = 4 * world
    nt_pad = (nt + unit - 1) // unit * unit
    groups = make_level_groups(offsets, 2, nt_pad, world, rank, spec)
    gx = GroupedExchange(groups, rank, world, torch.device("cpu"), comm_dtype=torch.bfloat16 if comm16 else None)
    base = torch.arange(nt_pad, dtype=torch.float32) % 1013
    grad = base * (rank + 1)                                                 # rank-dependent "gradient"; the mean is base * (world + 1) / 2
    grad[nt:] = 0
    store = torch.full((nt_pad,), float("nan"))
    fins = [gx.rs_async(g, grad) for g in groups]                            # all reduce-scatters in flight, like the trainer
    pend = []
    for g, fin in zip(groups, fins):
        fin()
        want = (base[g.lo:g.hi] * (world + 1) / 2.0)
        want[max(0, min(nt, g.hi) - g.lo):] = 0                              # (the padding behind the table)
        tol = dict(rtol=1e-2, atol=1e-2) if comm16 else dict(rtol=1e-6, atol=1e-4)
        torch.testing.assert_close(g.shard[:g.hi - g.lo], want, **tol)
        assert not grad[g.a:g.b].any()                                       # the local accumulator is cleared
        store[g.lo:g.hi] = -g.shard[:g.hi - g.lo] + rank * 0.0               # "the optimizer": a function of the averaged gradient
        pend.append(gx.ag_async(g, store))
    for fin in pend:
        fin()
    assert not torch.isnan(store).any()
    mine = store.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(torch.equal(both[0], x) for x in both[1:])                    # every rank ends with the same full table
    if not comm16:
        want = -(base * (world + 1) / 2.0); want[nt:] = 0
        torch.testing.assert_close(store, want, rtol=1e-6, atol=1e-4)
    m = torch.zeros(nt_pad)                                                  # checkpoint path: moments live on their owners only
    for g in groups:
        m[g.lo:g.hi] = float(rank + 1)
    gx.gather(m)
    for g in groups:
        for r in range(world):
            lo, hi = g.a + r * g.c, max(g.a + r * g.c, min(g.a + (r + 1) * g.c, g.b))
            assert bool((m[lo:hi] == float(r + 1)).all())
    open(os.path.join(out_dir, "ok_%d" % rank), "w").write("%s" % [gg.aligned for gg in groups])
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_exchange_eight_ranks_cpu(tmp_path):
    """The overlapped exchange's collectives on 8 ranks (the node size of BASELINE config 4), 3 groups, over gloo on CPU tensors:
    reduce-scatter = cross-rank mean of the own chunk, all-gather reassembles the table, also through the padded staging buffers
    (4 * world = 32 does not divide the coarse groups).  No GPU: the same GroupedExchange object FusedTrainer drives."""
    world = 8
    mp.spawn(_grouped_worker, args=(world, _free_port(), "6,3,0", False, str(tmp_path)), nprocs=world, join=True)
    oks = sorted(os.listdir(tmp_path))
    assert oks == ["ok_%d" % r for r in range(world)]
    assert "False" in open(os.path.join(tmp_path, "ok_0")).read()            # at least one group went through staging


def test_grouped_exchange_three_ranks_bf16_transport_cpu(tmp_path):
    world = 3
    mp.spawn(_grouped_worker, args=(world, _free_port(), "4,0", True, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok_%d" % r for r in range(world)]
