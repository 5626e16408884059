"""Ray-sharded data parallelism: every rank holds a full replica (hash table, MLPs, occupancy grid) and renders its own
shard of the ray batch (torch.distributed backend "nccl" == RCCL over xGMI on ROCm).  The reference has no multi-GPU path at
all; the gradient exchange is the only exchange step the hot path needs.  Two forms:
  * FusedTrainer (what bench.py --gpus N runs), default: reduce-scatter of the table gradient -> Adam on the own 1/N of the
    table -> all-gather of the updated parameters (`reduce_scatter_avg`, `all_gather_shards` below), plus one 37.6 KB all-reduce
    of [MLP gradient | inf flag]; `shard_optimizer=False`: ONE all-reduce of one flat bucket [table | MLP | flag];
  * GradReducer (the reference's loop shape: modules/ + torch.optim): the table gradient all-reduced in place + one small
    flattened bucket for the 9 408 MLP weights."""
import torch
import torch.distributed as dist


def shard_rays(n_total, rank, world):
    """Contiguous [start, stop) shard of a global batch of n_total rays for `rank` (remainder to the first ranks)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradReducer:
    """Averages .grad of every trainable parameter across ranks.

    Large tensors (the 45.7 MB hash-table gradient) are reduced in place with their own collective; everything
    else (9 408 MLP weights) travels as one flattened bucket.  MSE is a mean over the local shard, so gradients
    are averaged, not summed (equal shard sizes)."""

    def __init__(self, module, world=None, big_numel=1 << 20, group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = world if world is not None else dist.get_world_size(group)
        self.big_numel = big_numel
        self.group = group
        self._avg = None

    def _reduce(self, t):
        if self._avg is None:
            self._avg = dist.get_backend(self.group) == "nccl"
        if self._avg:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:                                   # gloo (CPU tests) has no AVG
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)

    @torch.no_grad()
    def all_reduce(self):
        if self.world == 1:
            return
        small = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)    # a rank whose shard produced no samples still takes part
            if p.numel() < self.big_numel:
                small.append(p)
            else:
                self._reduce(p.grad)
        if small:
            flat = torch.cat([p.grad.reshape(-1).float() for p in small])
            self._reduce(flat)
            off = 0
            for p in small:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n


def reduce_scatter_avg(out, inp, rank, world, group=None):
    """out <- average over ranks of shard `rank` of inp (inp = world equal shards).  RCCL: one reduce_scatter_tensor(AVG).
    Other backends (gloo: functional tests; it has no AVG and no CUDA reduce-scatter): all-reduce(SUM) + own shard / world."""
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=group)
        out.copy_(inp[rank * out.numel():(rank + 1) * out.numel()]).div_(world)
    return out


def all_gather_shards(store, rank, shard_len, world, group=None):
    """Every rank holds the valid shard [rank * shard_len, (rank + 1) * shard_len) of `store`: fill in everybody else's."""
    mine = store[rank * shard_len:(rank + 1) * shard_len]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(store, mine, group=group)        # in place: the own shard already sits at its offset
    else:
        dist.all_gather([store[r * shard_len:(r + 1) * shard_len] for r in range(world)], mine.clone(), group=group)
    return store


def broadcast_occupancy(model, src=0, group=None):
    """Keep replicas identical after an occupancy-grid update whose random cell sampling is rank-local."""
    dist.broadcast(model.density_grid, src=src, group=group)
    dist.broadcast(model.density_bitfield, src=src, group=group)
