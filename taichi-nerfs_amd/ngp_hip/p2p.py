"""Direct gradient exchange over peer memory (csrc/exchange.hip) -- the prototype of SURVEY section 8(e)'s "direct (all-links)
reduce-scatter + all-gather" (VERDICT r5 item 7b).

`PeerExchange` gives FusedTrainer the two operations its sharded optimizer needs, with the collectives replaced by ONE push launch and
ONE wait launch each:
    reduce_scatter_avg(out, inp)   every rank writes slice p of its padded table gradient into row `rank` of rank p's inbox, raises a flag
                                   there; rank p waits for its `world` flags and reduces the rows to the average of its shard
    all_gather(store, sl)          every rank writes its updated shard into every peer's copy of `store` at the same offset + flag; wait
Peers' buffers are reached through hipIpc mappings: the inbox, the flag words and the gathered stores are allocated with torch, their IPC
handles travel once through torch.distributed (all_gather_object), and every rank opens every other rank's.  Flags carry the step number,
so nothing is ever reset; a step's all-gather is the barrier that keeps a fast rank from overwriting an inbox row that is still being read.
UNMEASURED on hardware: exercised by two and three processes on ONE device (tests/test_gpu_p2p.py) -- every peer pointer there is a real
IPC mapping of another process's memory, but no byte has crossed xGMI.  Selected with FusedTrainer(exchange="p2p") / bench.py --comm-path p2p."""
import ctypes

import torch
import torch.distributed as dist

from . import lib as _lib_mod
from .lib import check
from .ops import _ptr, _stream


def _share(t):
    """IPC handle of a CUDA tensor's storage (torch's own reduction: hipIpcGetMemHandle on the allocation + the offset inside it)."""
    return (t.untyped_storage()._share_cuda_(), t.storage_offset() * t.element_size(), t.numel(), str(t.dtype))


def _open(handle, device):
    meta, byte_off, numel, dtype = handle
    dtype = getattr(torch, dtype.split(".")[1])
    storage = torch.UntypedStorage._new_shared_cuda(*meta)
    t = torch.empty(0, dtype=dtype, device=device)
    t.set_(storage, byte_off // t.element_size(), (numel,))
    return t


class PeerExchange:
    MAX_SPINS = 1 << 22                  # polls of the flag words per block before a wait gives up (~0.1-1 s: a hang must not hang the box)

    def __init__(self, rank, world, device, shard_len, stores, group=None):
        """stores: the full-size buffers the all-gather fills ({name: tensor of world * shard_len elements}, e.g. the table / its 16-bit copy)."""
        L = self.L = _lib_mod.load()
        if world > L.ngp_p2p_max_peers():
            raise ValueError("PeerExchange supports up to %d ranks" % L.ngp_p2p_max_peers())
        self.rank, self.world, self.device, self.shard_len, self.group = rank, world, device, shard_len, group
        self.inbox = torch.zeros(world, shard_len, device=device, dtype=torch.float32)
        self.flags = torch.zeros(2, world, device=device, dtype=torch.int32)       # [0] reduce-scatter arrivals, [1] all-gather arrivals
        self.done = torch.zeros(4, device=device, dtype=torch.int32)
        self.err = torch.zeros(1, device=device, dtype=torch.int32)
        self.step = 0
        self.stores = dict(stores)
        mine = {"inbox": _share(self.inbox), "flags": _share(self.flags)}
        mine.update({"store:" + k: _share(v) for k, v in self.stores.items()})
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        self._keep = []                     # the opened mappings stay alive as long as this object
        self.peer = []
        for r in range(world):
            if r == rank:
                ent = {"inbox": self.inbox, "flags": self.flags}
                ent.update({"store:" + k: v for k, v in self.stores.items()})
            else:
                ent = {k: _open(h, device) for k, h in everyone[r].items()}
            self._keep.append(ent)
            self.peer.append(ent)
        P = ctypes.c_void_p * world
        # row `rank` of every peer's inbox; the peers' flag rows; the peers' stores
        self._dst_inbox = P(*[ctypes.c_void_p(self.peer[r]["inbox"].data_ptr() + rank * shard_len * 4) for r in range(world)])
        self._flag_rs = P(*[ctypes.c_void_p(self.peer[r]["flags"].data_ptr()) for r in range(world)])
        self._flag_ag = P(*[ctypes.c_void_p(self.peer[r]["flags"].data_ptr() + world * 4) for r in range(world)])
        self._dst_store = {k: P(*[ctypes.c_void_p(self.peer[r]["store:" + k].data_ptr()) for r in range(world)]) for k in self.stores}
        dist.barrier(group=group)           # nobody pushes before everybody has mapped everybody

    def reduce_scatter_avg(self, out, inp):
        """out [shard_len] f32 <- average over ranks of slice `rank` of inp [world * shard_len] f32."""
        assert inp.dtype == torch.float32 and inp.numel() == self.world * self.shard_len and out.numel() == self.shard_len
        self.step += 1
        st = _stream()
        check(self.L.ngp_p2p_push(_ptr(inp), self.shard_len, 4, self.world, self._dst_inbox, self._flag_rs, 0, 0, self.rank, self.step,
                                  ctypes.c_void_p(self.done.data_ptr()), st), "ngp_p2p_push")
        check(self.L.ngp_p2p_wait(ctypes.c_void_p(self.flags.data_ptr()), self.world, self.step, self.MAX_SPINS, _ptr(self.inbox), self.shard_len,
                                  1.0 / self.world, 1, _ptr(out), _ptr(self.err), st), "ngp_p2p_wait")
        return out

    def all_gather(self, name, sl):
        """Every rank's shard store[sl] (its own `rank`-th shard) lands in every peer's store at the same offset."""
        store = self.stores[name]
        st = _stream()
        mine = store[sl]
        check(self.L.ngp_p2p_push(_ptr(mine), mine.numel(), store.element_size(), self.world, self._dst_store[name], self._flag_ag, sl.start, 1,
                                  self.rank, self.step, ctypes.c_void_p(self.done.data_ptr() + 4), st), "ngp_p2p_push")
        check(self.L.ngp_p2p_wait(ctypes.c_void_p(self.flags.data_ptr() + 4 * self.world), self.world, self.step, self.MAX_SPINS, _ptr(None), 0,
                                  0.0, 0, _ptr(None), _ptr(self.err), st), "ngp_p2p_wait")
        return store

    def check_errors(self):
        """Host sync: raises if a wait ran out of spins (a peer that never arrived)."""
        if int(self.err.item()) != 0:
            raise RuntimeError("PeerExchange: a wait for the peers' flags timed out on rank %d (step %d)" % (self.rank, self.step))
