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


# ---- overlapped exchange: the table in LEVEL GROUPS, every rank owning the rank-th 1/world of each group ------------------------
class LevelGroup:
    """Levels [l0, l1) of the hash table = flat float range [a, b) of the (padded) table storage; rank r owns the r-th chunk of
    c = ceil((b - a) / (4 world)) * 4 floats, clipped to the group: [lo, hi)."""
    __slots__ = ("index", "l0", "l1", "mask", "a", "b", "c", "aligned", "lo", "hi", "shard", "stage", "comm", "comm_shard")


def make_level_groups(offsets, n_features, nt_pad, world, rank, spec):
    """offsets[l] = first ENTRY of level l (ngp_hash_levels.offset), spec = first level of each group in launch order (strictly
    descending, ending in 0, e.g. "8,0").  The first group ends at the last level and also takes the padding behind the table
    ([.., nt_pad)).  A group whose length is not a multiple of 4 * world is exchanged through zero-padded staging buffers."""
    n_levels = len(offsets)
    starts = [int(x) for x in str(spec).split(",") if x.strip() != ""]
    if not starts or starts[-1] != 0 or any(a <= b for a, b in zip(starts, starts[1:])) or starts[0] >= n_levels:
        raise ValueError("level groups must be strictly descending first levels ending in 0, e.g. '12,8,0' (got %r)" % (spec,))
    groups, hi_level = [], n_levels
    for k, l0 in enumerate(starts):
        g = LevelGroup()
        g.index, g.l0, g.l1 = k, l0, hi_level
        g.mask = sum(1 << l for l in range(l0, hi_level))
        g.a = n_features * int(offsets[l0])
        g.b = int(nt_pad) if k == 0 else n_features * int(offsets[hi_level])
        n = g.b - g.a
        g.c = (n + 4 * world - 1) // (4 * world) * 4
        g.aligned = g.c * world == n
        g.lo = g.a + rank * g.c
        g.hi = max(g.lo, min(g.lo + g.c, g.b))
        g.shard, g.stage, g.comm, g.comm_shard = None, {}, None, None
        groups.append(g)
        hi_level = l0
    return groups


class GroupedExchange:
    """Reduce-scatter (cross-rank average) and all-gather of the table, group by group, with asynchronous collectives: `rs_async`
    / `ag_async` start one and return `finish()`, which makes the current stream wait for it and does the local post-processing.
    Device-agnostic (RCCL on GPUs; gloo on CPU tensors in tests/test_dist_gloo.py).  `stub` replaces every collective by its
    local part (bench.py: the step without communication); `timed(name, fn)` lets the caller bracket the waits."""

    def __init__(self, groups, rank, world, device, group=None, comm_dtype=None, timed=None):
        self.groups, self.rank, self.world, self.device, self.group = groups, rank, world, device, group
        self.stub = False
        self.timed = timed or (lambda name, fn: fn())
        for g in groups:
            g.shard = torch.zeros(g.c, device=device, dtype=torch.float32)
            if comm_dtype is not None:
                g.comm = torch.empty(g.c * world, device=device, dtype=comm_dtype)
                g.comm_shard = torch.empty(g.c, device=device, dtype=comm_dtype)

    def _nccl(self):
        return dist.get_backend(self.group) == "nccl"

    def _stage(self, g, dtype):
        buf = g.stage.get(dtype)
        if buf is None:
            buf = g.stage[dtype] = torch.zeros(g.c * self.world, device=self.device, dtype=dtype)
        return buf

    def rs_async(self, g, src):
        """Start the reduce-scatter of src[g.a:g.b] (flat gradient accumulator); finish() leaves the average of this rank's chunk
        in g.shard and clears src[g.a:g.b] for the next step."""
        n = g.b - g.a
        inp = src[g.a:g.b]
        if not g.aligned:
            st_ = self._stage(g, torch.float32)
            st_[:n].copy_(inp)
            inp = st_
        out = g.shard
        if g.comm is not None:
            g.comm.copy_(inp)
            inp, out = g.comm, g.comm_shard
        rank, world, c = self.rank, self.world, g.c
        if self.stub:
            work, post = None, (lambda: out.copy_(inp[rank * c:(rank + 1) * c]))
        elif self._nccl():
            work, post = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None
        else:                                      # gloo (functional tests): no AVG, no reduce-scatter on its CUDA path
            work = dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            post = lambda: out.copy_(inp[rank * c:(rank + 1) * c]).div_(world)

        def finish():
            def wait():
                if work is not None:
                    work.wait()
            self.timed("wait_reduce_scatter_group%d" % g.index, wait)
            if post is not None:
                post()
            if out is not g.shard:
                g.shard.copy_(out)
            src[g.a:g.b].zero_()
        return finish

    def ag_async(self, g, store):
        """Start the all-gather of group g's part of `store` (every rank holds its own chunk [g.lo, g.hi)); returns finish()."""
        n = g.b - g.a
        if self.stub:
            return lambda: None
        world, c = self.world, g.c
        if g.aligned:
            mine = store[g.lo:g.lo + c]
            if self._nccl():
                work = dist.all_gather_into_tensor(store[g.a:g.b], mine, group=self.group, async_op=True)
            else:
                work = dist.all_gather([store[g.a + r * c:g.a + (r + 1) * c] for r in range(world)], mine.clone(), group=self.group,
                                       async_op=True)
            post = None
        else:
            st_ = self._stage(g, store.dtype)
            mine = st_[self.rank * c:(self.rank + 1) * c]
            mine[:g.hi - g.lo].copy_(store[g.lo:g.hi])
            if self._nccl():
                work = dist.all_gather_into_tensor(st_, mine, group=self.group, async_op=True)
            else:
                work = dist.all_gather([st_[r * c:(r + 1) * c] for r in range(world)], mine.clone(), group=self.group, async_op=True)
            post = lambda: store[g.a:g.b].copy_(st_[:n])

        def finish():
            self.timed("wait_all_gather_group%d" % g.index, work.wait)
            if post is not None:
                post()
        return finish

    def gather(self, store):
        """All-gather a full-size buffer group by group, synchronously (checkpoints, the fp32 master behind a 16-bit copy)."""
        for g in self.groups:
            self.ag_async(g, store)()


def broadcast_occupancy(model, src=0, group=None):
    """Keep replicas identical after an occupancy-grid update whose random cell sampling is rank-local."""
    dist.broadcast(model.density_grid, src=src, group=group)
    dist.broadcast(model.density_bitfield, src=src, group=group)
