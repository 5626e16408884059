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
