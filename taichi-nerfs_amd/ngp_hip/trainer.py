"""FusedTrainer -- one Instant-NGP optimisation step as a fixed sequence of 12 kernel launches, no host sync.

It performs exactly what the reference's training iteration does (train.py:168-201) for the default model (fp32, bf16-copy or
half2 hash encoder):
    rays -> render (march, hash encode, MLPs, composite) -> MSE [+ distortion] vs target -> backward -> GradScaler
    -> Adam(eps=1e-15) -> CosineAnnealingLR
but every piece is a libngp_hip kernel working on persistent buffers:
  * gradients are accumulated by the backward kernels straight into persistent buffers (no per-step 45.7 MB allocation +
    memset, no autograd graph), and are unscaled + zeroed inside the Adam pass;
  * the backward runs over the LIVE samples only -- those in front of each ray's early-termination point, the others have
    exact-zero gradients -- through a compacted index list (ngp_live_compact);
  * the inf/nan check, loss-scale growth/backoff, learning rate and bias corrections live in a tiny device-side state
    (ngp_train_prologue), so nothing is read back; table Adam, MLP Adam and the fp16 fragment repack are one launch;
  * the NEXT batch's march runs on a side stream underneath the current step (it only depends on rays and the bitfield);
  * with world_size > 1 each rank renders its own ray shard and the gradient bucket (table | MLP | inf flag) is all-reduced
    (one collective over one flat bucket) over RCCL before the optimizer kernels -- the only exchange step.
`capture()` switches the shading -> optimizer chain to hipGraph replay (measured slower than eager on ROCm 7: not the default).

The reference's own loop (torch.optim.Adam + torch GradScaler + autograd through modules/) keeps working on the drop-in
operators; this class is the MI355X-native fast path that bench.py measures."""
import ctypes
import math
import os

import torch
import torch.distributed as dist

from . import experiment as _exp
from . import lib as _lib_mod
from .fused import RenderConfig, TrainArena
from .lib import check
from .ops import MLP_N_WEIGHTS, _ptr, _stream

_SF_LOSS_SCALE, _SF_LOSS = 0, 5
_SI_ITER, _SI_OPT_STEP, _SI_FOUND_INF, _SI_SKIPPED = 0, 1, 3, 5


class FusedTrainer:
    _MARCH_NARROW_MAX = 1_000_000          # marched samples per step up to which the prefetched march goes to the start of the step

    @staticmethod
    def chunk_rounds(max_samples):
        """Rounds of the chunked forward as (begin, length, previous begin): 64, 64, 128, 256, 512, ... samples of every live ray,
        every boundary a multiple of 64 (the compositing kernels read a ray 64 samples at a time), together covering max_samples."""
        rounds, b, l, pb = [], 0, 64, 0
        while b < max_samples:
            l = min(l, max_samples - b)
            rounds.append((b, l, pb))
            pb, b = b, b + l
            l = b
        return rounds

    def __init__(self, model, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, max_steps=20000, eta_min=None, init_scale=2.0**19,
                 growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, exp_step_factor=0.0, T_threshold=1e-4,
                 max_samples=1024, process_group=None, world_size=None, grad_comm_dtype=None,
                 distortion_loss_w=0.0, shard_optimizer=None, chunked_forward=None, exchange="rccl"):
        if not model.use_fused_mlp:
            raise ValueError("FusedTrainer needs the default architecture (L=16, F=2 hash grid, 64-wide MLPs)")
        self.half = bool(model.half_opt)              # half2 encoder (hash_encoder_half.py): f16 table copy, f16 gradient buffer
        self.model = model
        self.L = _lib_mod.load()
        dev = model.pos_encoder.hash_table.device
        if dev.type != "cuda":
            raise RuntimeError("FusedTrainer needs the model on a GPU (libngp_hip has no CPU path)")
        self.dev = dev
        self.lr0, self.eta_min = float(lr), float(lr / 30 if eta_min is None else eta_min)
        self.t_max = int(max_steps)
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.growth, self.backoff, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.exp_step_factor, self.T_threshold, self.max_samples = float(exp_step_factor), float(T_threshold), int(max_samples)
        self.bg = 1.0 if exp_step_factor == 0 else 0.0                    # rendering.py:219-226
        self.distortion_loss_w = float(distortion_loss_w)                # train.py:194-195 (0 = off, the reference default)
        self._init_options(model, exp_step_factor, max_samples, chunked_forward, process_group, world_size)
        self._init_buffers(model, dev, init_scale, grad_comm_dtype, shard_optimizer, process_group)
        self._init_prefetch(dev)
        self._init_table_copies(model, dev, exp_step_factor)
        self._init_exchange(model, dev, exchange)
        self.repack()

    # ---- construction, piece by piece (round 6: __init__ was one 260-line body) ------------------------------------------------
    def _init_options(self, model, exp_step_factor, max_samples, chunked_forward, process_group, world_size):
        """Launch-sequence options: chunked forward, live backward, which march, which scatter-add; the process group."""
        # Round 5 -- chunked forward: shade a ray's samples in growing chunks (64, 64, 128, 256, ...) and stop at the chunk in which
        # its transmittance falls to the compositing threshold, instead of shading everything the march emitted (the reference shades
        # all of it and ignores what lies behind T <= 1e-4, volume_train.py:38; on the C3 shape that is 3 of 4 samples).  Results per
        # ray and all gradients are unchanged (tests/test_gpu_chunked.py); `rm_samples` stays the MARCHED count, shaded_samples()
        # reports what was shaded.  Default: on for multi-cascade / exponentially stepped scenes (where rays run far past their
        # surfaces), off for the bounded synthetic ones (C2: 45 marched vs 43 composited samples per ray -- five rounds of three
        # launches would cost more than they save); chunked_forward=True / False or NGP_EXPERIMENT chunked_fwd=1 / 0 override.
        ck = _exp.get("chunked_fwd")
        if chunked_forward is None and ck is not None:
            chunked_forward = ck == "1"
        if chunked_forward is None:
            chunked_forward = float(exp_step_factor) > 0 or int(model.cascades) > 1
        self.chunked = bool(chunked_forward) and not self.half and int(max_samples) % 64 == 0 and int(max_samples) >= 128
        self._chunk_rounds = self.chunk_rounds(int(max_samples)) if self.chunked else []
        self._chunk_counts = None                      # [2, rounds] int32: list lengths per round, one set per step parity
        self._chunk_T = {}
        # backward over the samples in front of each ray's early-termination point only (NGP_EXPERIMENT live_backward=0: over all of them)
        self.live_backward = _exp.get("live_backward", "1") != "0"
        self.march_fused = _exp.get("march_fused", "1") != "0"
        self._march_rng = _exp.get("march_rng", "kernel") != "torch"     # torch: a torch.rand vector per march (rounds 1-3)
        # table gradient (fp32, or fp16 for the half2 encoder): "sliced" = LDS-owned table slices, no global float atomics
        # (csrc/hash_bwd_lds.hip; the default whenever the level table fits: F = 2, levels of <= 2^19 entries), "atomic" = round
        # 1's float-atomic / packed-f16-atomic kernels
        self.hash_bwd = os.environ.get("NGP_HASH_BWD", "sliced")
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)

    def _init_buffers(self, model, dev, init_scale, grad_comm_dtype, shard_optimizer, process_group):
        """Parameters as views of flat / padded storage, gradient buckets, Adam moments, the device-side step state."""

        # the five MLP weights become views of one flat buffer so their gradient is the kernel's flat dW
        ws = list(model._mlp_weights())
        flat = torch.cat([w.detach().reshape(-1) for w in ws]).contiguous()
        assert flat.numel() == MLP_N_WEIGHTS
        off = 0
        for w in ws:
            n = w.numel()
            w.data = flat[off:off + n].view_as(w)
            off += n
        self.mlp_flat = flat
        f32 = dict(device=dev, dtype=torch.float32)
        nt = model.pos_encoder.hash_table.numel()
        assert nt % 8 == 0
        # world > 1, sharded optimizer (the default): the table gradient is reduce-scattered, every rank runs Adam on ITS 1/world
        # of the table (the dense 45.7 MB pass shrinks by `world`) and the updated parameters are all-gathered -- as the 16-bit
        # copy the forward reads when there is one (bf16 / half2 encoder: half the bytes on xGMI).  Same bytes on the wire as
        # one all-reduce (which IS reduce-scatter + all-gather), no replicated optimizer work.  shard_optimizer=False keeps
        # round 1's single all-reduce of one flat bucket + replicated Adam.
        # (shard_optimizer=True with world_size 1 runs the same collectives over a 1-rank group: used to exercise them on RCCL)
        self.shard = bool(self.world > 1 if shard_optimizer is None else shard_optimizer)
        if self.shard and not dist.is_initialized():
            raise RuntimeError("the sharded optimizer needs an initialised torch.distributed process group")
        self.rank = dist.get_rank(process_group) if (self.shard or self.world > 1) and dist.is_initialized() else 0
        # shards are whole float4 groups of equal size: parameter, gradient and 16-bit copies live in storage padded to a multiple
        # of 4 * world elements (the parameters become views of it, exactly like the five MLP weights above)
        unit = 4 * self.world if self.shard else 4
        self.nt, self.nt_pad = nt, (nt + unit - 1) // unit * unit
        self.shard_len = self.nt_pad // self.world if self.shard else self.nt_pad
        hp = model.pos_encoder.hash_table
        store = torch.zeros(self.nt_pad, **f32)
        store[:nt].copy_(hp.detach().reshape(-1))
        hp.data = store[:nt].view_as(hp)
        self.table_store = store
        self.table = store[:nt]                                          # [entries * 2] (the half encoder's parameter is 2-D)
        # gradient bucket(s).  Unsharded: ONE flat bucket [hash-table grad | MLP grad | inf flag] = a single all-reduce per step.
        # Sharded: [padded table grad] is reduce-scattered, [MLP grad | inf flag] (37.6 KB) is all-reduced.
        if self.half:
            # the half2 encoder accumulates its table gradient in f16 (one packed atomic per corner): separate f16 buffer
            self.table_grad_store = torch.zeros(self.nt_pad, device=dev, dtype=torch.float16)
            self.table_grad = self.table_grad_store[:nt]
            self.grad_flat = torch.zeros(MLP_N_WEIGHTS + 4, **f32)
            self.mlp_grad = self.grad_flat[:MLP_N_WEIGHTS]
            self._flag_f = self.grad_flat[MLP_N_WEIGHTS:MLP_N_WEIGHTS + 1]
            self.small_bucket = self.grad_flat
        else:
            self.grad_flat = torch.zeros(self.nt_pad + MLP_N_WEIGHTS + 4, **f32)
            self.table_grad_store = self.grad_flat[:self.nt_pad]
            self.table_grad = self.grad_flat[:nt]
            self.mlp_grad = self.grad_flat[self.nt_pad:self.nt_pad + MLP_N_WEIGHTS]
            self._flag_f = self.grad_flat[self.nt_pad + MLP_N_WEIGHTS:self.nt_pad + MLP_N_WEIGHTS + 1]
            self.small_bucket = self.grad_flat[self.nt_pad:]
        self.shard_grad = (torch.zeros(self.shard_len, device=dev, dtype=self.table_grad_store.dtype) if self.shard else None)
        # optional 16-bit gradient transport: halves the bytes on xGMI; fp32 (exact mean) is the default
        # gradient transport between ranks.  None (default, round 6): bf16 when the model ALREADY reads a bf16 storage copy of its table
        # (NGP(table_dtype=torch.bfloat16): its forward sees 8 bits of mantissa per parameter anyway, and the updated parameters
        # travel back as that copy) -- half the bytes in both directions of the exchange, 83 % against 68 % modelled scaling efficiency
        # at 300 GB/s; fp32 (the exact mean) otherwise.  Pass torch.float32 / torch.bfloat16 to pin it.
        if grad_comm_dtype is None:
            enc_dt = getattr(model.pos_encoder, "table_dtype", torch.float32)
            grad_comm_dtype = torch.bfloat16 if (enc_dt == torch.bfloat16 and not self.half) else torch.float32
        self.grad_comm_dtype = grad_comm_dtype
        if grad_comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_comm_dtype must be torch.float32 or torch.bfloat16")
        self._comm = self._comm_shard = None
        if grad_comm_dtype != torch.float32 and self.world > 1 and not self.half:
            if self.shard:
                self._comm = torch.empty(self.nt_pad, device=dev, dtype=grad_comm_dtype)
                self._comm_shard = torch.empty(self.shard_len, device=dev, dtype=grad_comm_dtype)
            else:
                self._comm = torch.empty_like(self.grad_flat, dtype=grad_comm_dtype)
        # per-block slabs of the MLP backward's weight gradients (ngp_mlp_bwd_live_parts), summed by the prologue launch
        self.mlp_parts = torch.empty(self.L.ngp_mlp_dw_parts_max() * MLP_N_WEIGHTS, **f32)
        self._dw_atomic = _exp.get("mlp_dw", "") == "atomic"
        self.table_m, self.table_v = torch.zeros(self.nt_pad, **f32), torch.zeros(self.nt_pad, **f32)
        self.mlp_m, self.mlp_v = torch.zeros(MLP_N_WEIGHTS, **f32), torch.zeros(MLP_N_WEIGHTS, **f32)
        self.state_f = torch.zeros(8, **f32)
        self.state_i = torch.zeros(8, device=dev, dtype=torch.int32)
        self.state_f[_SF_LOSS_SCALE] = float(init_scale)
        # live-sample counters, one per step parity: the fused composite kernel appends to the current step's counter (which has
        # to be 0 at launch) and clears the other one for the next step -- no memset launch on the step's critical path
        self._live_pair = torch.zeros(2, device=dev, dtype=torch.int32)    # (parity = which of the two march sets the step shades)
        self._graph = None
        self._launch_incomplete = False       # a _launch() that raised between flipping the march-set parity and issuing the step
        self._grads_only = False
        self.stats = {}
        self._sets = {}
        self._cur = 0

    def _init_prefetch(self, dev):
        """The side stream(s) the next batch's march runs on, and where in the step it is issued."""
        self._side = torch.cuda.Stream(device=dev)
        # The side stream runs at the device's LOWEST priority (NGP_EXPERIMENT side_priority=default: torch's): since the march draws its jitter
        # itself (no torch uniform_ kernel in front of it any more) it is ready the moment the scatter-add is launched, and at equal
        # priority its blocks were dispatched in front of the scatter-add's persistent workgroups and ran beside them for the whole
        # launch (scatter-add 190 -> 210 us, march 235 us: profiles/r04_rocprofv3_timed_region_equal_priority.txt); at low priority
        # it takes the CUs the scatter-add's workgroups leave as they retire (A/B on one box, 2 x 200 steps each: 0.882-0.894 vs
        # 0.911-0.923 us per 1000 live samples)
        # Round 5 (one GPU, fp32 master table): the table's optimizer moved into the scatter-add, the HBM-bound launch the march used to
        # hide under is gone, and a 16-wave-per-CU march under the scatter-add now outlives the step (its blocks only get CUs as the
        # persistent workgroups retire: profiles/r05_rocprofv3_timed_region_march_under_scatter.txt, the next step's gather waits
        # 21 us for it).  profiles/r05_march_placement.txt: issued at the START of the step as 4-wave blocks on a default-priority
        # stream it costs least (1.287 vs 1.305 ns per live sample under the scatter-add, 1.323 in line).
        # (C3 shape -- six cascades, exponential stepping -- is the other way round: the march is 1.3 ms of ALU work and the scatter-add
        # 2.3 ms, so round 4's arrangement, 16-wave blocks at low priority under the scatter-add, hides it best: 13.5 M rays/s against
        # 12.4 at the start of the step, profiles/r05_bench_garden_c3_march_placement.txt.)
        # Which of the two is right depends on how heavy the march is, and that changes during a training run (a young model's occupancy
        # grid is dense: 260 samples per ray; a trained one's is sparse: 45).  Measured on one box, same round: C2 (8192 rays, 376 k
        # marched samples) 17.1 M rays/s with the narrow march at the start of the step against 16.6 under the scatter-add; the
        # initialisation regime (2.1 M marched) 3.5 M against 4.0; 65 536 rays per step (2.8 M marched) 20.9 against 21.9.  So the
        # trainer ADAPTS: the side stream copies each prefetched march's sample count to pinned host memory (asynchronously: nothing
        # waits for it) and the next hooks read whatever has arrived -- at most a few steps old.  Up to _MARCH_NARROW_MAX marched
        # samples: start of the step, 4-wave blocks, default priority; above: before the scatter-add at low priority -- as 16-wave
        # blocks until the end of round 5, when 4-wave blocks measured better or equal in the three heavy cases (C3 with the
        # concentrated-scene scatter-add 15.2-15.5 M rays/s against 14.6, 65 536 Lego rays 31.0-31.2 against 29.6, the initialisation
        # regime 4.07 against 4.08: profiles/r05_bench_garden_c3_concentrated.txt, r05_heavy_march_placement.txt).
        # Any of NGP_EXPERIMENT prefetch_at / NGP_EXPERIMENT march_shape / NGP_EXPERIMENT side_priority pins the arrangement instead.
        self._one_gpu_flush = (self.world == 1 and not self.half and _exp.get("flush_adam", "1") != "0"
                               and self.hash_bwd == "sliced")
        pinned = any(_exp.has(k) for k in ("prefetch_at", "march_shape", "side_priority"))
        self._adaptive_prefetch = self._one_gpu_flush and not pinned
        self._host_wait_ok = _exp.get("prefetch_host_wait", "0") == "1"       # opt-in: see step()
        self._side_prio = None
        self._side_default = self._side
        self._side_low = None
        if self._adaptive_prefetch or _exp.get("side_priority", "low") == "low":
            h, lo, hi = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
            check(self.L.ngp_stream_create_low_priority(ctypes.byref(h), ctypes.byref(lo), ctypes.byref(hi)), "ngp_stream_create_low_priority")
            self._side_low = torch.cuda.ExternalStream(h.value, device=dev)
            self._side_prio = (lo.value, hi.value)
            self._side = self._side_low
        self._hook_at = 3
        self._marched_host = None              # pinned host int32 (ngp_host_alloc): sample count of a recent prefetched march
        if self._adaptive_prefetch:
            h = ctypes.c_void_p()
            check(self.L.ngp_host_alloc(ctypes.byref(h), 64), "ngp_host_alloc")
            self._marched_host = h
            ctypes.c_int32.from_address(h.value).value = 1 << 30          # (unknown yet: assume a young, dense model)
        self._ev_start = self._DevEvent(self.L)              # main stream -> side stream: the prefetch may start
        self.prefetch_hits = 0                # steps that consumed a march prefetched by the previous step() call
        # where in the step the next batch's march is put on the side stream: 0 = at the start, 1 = after the hash gather
        # (default: with the backward running on the live samples only, the ~115 us march chain has to start this early to be
        # done before the step is; A/B on one box: 0.327 ms at 1, 0.331 at 0, 0.342 at 2; round 2: 0.567-0.571 at 1, 0.578-0.583 at
        # 0, 0.585 at 2, 0.590 at 3), 2 = after the MLP forward, 3 = before the scatter-add, 4 = after the scatter-add.
        # Round 3: the count kernel's replay became parallel (109 -> 47 us, ~75 us chain), so the chain no longer has to start that
        # early and is best kept off the gather-bound encoder: at 350 k live samples, 3 runs x 400 steps each: 0.554 ms at 2,
        # 0.556 at 3, 0.560 at 0, 0.564 at 1 (not prefetched: 0.570).  Then the march became ONE launch (ngp_march_train_fused,
        # ~55 us) and the live list a by-product of the composite kernel; 3 runs x 400 steps each, no per-kernel events, at
        # 350 k live: 0.525 ms at 3 (under the scatter-add), 0.532 at 0, 0.538 at 2, 0.553 at 1, not prefetched 0.548
        import os as _os
        # (with world > 1 the default is 4: the march then runs underneath the gradient exchange -- RCCL's kernels occupy a few
        # workgroups and wait on xGMI -- instead of competing with the VALU-bound kernels of the step for issue slots)
        self._prefetch_at = float(_exp.get("prefetch_at", "3" if self.world == 1 else "4"))
        # Round 5: the SHAPE of the prefetched launch (ngp_march_train_fused_shaped): "waves per block, idle LDS bytes per block",
        # e.g. "4,82944" = 4-wave blocks, at most one per CU.  With the table's optimizer inside the scatter-add there is no
        # HBM-bound launch left to hide a 16-wave-per-CU march under; a narrow march asks every CU for one wave slot per SIMD and
        # runs beside whatever the step is doing.  Unset (and "16,0"): the 16-wave block every other march launch uses -- except
        # that on one GPU with the optimizer in the flush the trainer chooses per step (see _adaptive_prefetch above).
        shape = _exp.get("march_shape", "")
        self._march_shape = tuple(int(x) for x in shape.split(",")) if shape and shape != "16,0" else None
        self.sync_occupancy = True            # world > 1: broadcast rank 0's occupancy after every update_density_grid()
        # bench.py: callable -> (record this step?, list); every collective of the step is then bracketed by two HIP events on the
        # step's stream and (name, e0, e1) is appended to the list
        self.comm_probe = None

    def _init_table_copies(self, model, dev, exp_step_factor):
        """The fp16 MFMA weight image, the 16-bit storage copy of the table (bf16 copy / half2 encoder), the scatter-add's plan modes."""
        nt = self.nt
        lvs = model.pos_encoder.levels_struct
        self.enc_pairs = 1 if (lvs.n_levels == 16 and lvs.n_features == 2) else 0
        self.wpack = torch.empty(self.L.ngp_mlp_wpack_halfs(), device=dev, dtype=torch.float16)
        # bf16 storage copy of the table (HashEncoder(table_dtype=torch.bfloat16)): gathered by the forward, refreshed by Adam
        enc = model.pos_encoder
        self.table_bf16 = self.table_f16 = self.copy16_store = None
        if getattr(enc, "table_dtype", torch.float32) == torch.bfloat16:
            self.copy16_store = torch.zeros(self.nt_pad, device=dev, dtype=torch.bfloat16)
            enc._bf16, enc._bf16_ver = self.copy16_store[:nt].view(enc.hash_table.shape), None     # the encoder's copy IS this view
            self.table_bf16 = enc.table_bf16()
        elif self.half:
            self.copy16_store = torch.zeros(self.nt_pad, device=dev, dtype=torch.float16)
            enc._f16, enc._f16_ver = self.copy16_store[:nt].view(enc.hash_table.shape), None
            self.table_f16 = enc.table_f16().view(-1)
        self._master_stale = False            # sharded + 16-bit copy: the fp32 master of the other ranks' shards is gathered lazily
        # NGP_EXPERIMENT comm_overlap=1 (sharded exchange, fp32 gradient, LDS-sliced scatter-add): the scatter-add is issued as one launch per
        # LEVEL GROUP (NGP_EXPERIMENT comm_groups = first level of each group in launch order, default "8,0": levels 8-15, then 0-7) and a
        # group's reduce-scatter travels while the next group is still being accumulated; every rank owns the rank-th 1/world of
        # EACH group.  Default off until a multi-GPU run has decided.  NGP_EXPERIMENT comm_stub=1 replaces every collective
        # by its local part (bench.py: the step without communication, i.e. what of comm_ms is exposed).
        # Round 5, one GPU: the table's optimizer rides in the scatter-add's flush (ngp_hash_bwd_sliced_main_adam: the owner of a
        # non-replicated slice applies Adam to its 8192 entries instead of writing their gradient out for another launch to read
        # back); the optimizer launch shrinks to the replicated coarse levels [0, _adam_prefix) + the MLP.  The GradScaler decision
        # then has to exist before the scatter-add: the prologue moves in front of it (the MLP backward raises the inf flag on the
        # same d_enc values the scatter-add would).  NGP_EXPERIMENT flush_adam=0: the two-launch path (bit-identical results).
        self._flush_adam = _exp.get("flush_adam", "1") != "0"
        self._fold_prologue = True            # ... and the step's scalar bookkeeping inside that launch (ngp_hash_bwd_sliced_main_adam_step)
        self._adam_prefix = {}                # per scatter-add mode: floats of the table the flush does NOT update (-2: not expressible)
        # Deterministic mode (set_deterministic / NGP_DETERMINISTIC=1; bench.py conditions its model in it so that two processes
        # reach the same state): rays packed in ray order (count / scan / write chain), the live list in ray order
        # (ngp_live_compact), every table slice owned by one workgroup (NGP_BWD_PLAN_DETERMINISTIC: no float atomics), the
        # occupancy update without its two order-dependent spots (ngp_hip/occupancy.py).  Same kernels, same arithmetic per sample;
        # what changes is the ORDER in which floating-point sums are formed, which is fixed.  Slower (~1.5 ms per step at C2).
        self.deterministic = False
        # Multi-cascade / exponentially stepped scenes fill a small part of their box: the scatter-add's plan then treats the coarse
        # hashed levels like dense ones (NGP_BWD_PLAN_CONCENTRATED; C3: the launch 2.5 -> 1.7 ms beside the march, which then no
        # longer fits under it as 16-wave blocks: 4-wave blocks, profiles/r05_bench_garden_c3_concentrated.txt).  NGP_EXPERIMENT bwd_concentrated=0 / 1 overrides.
        conc = _exp.get("bwd_concentrated")
        self._concentrated = (conc == "1") if conc is not None else (float(exp_step_factor) > 0 or int(model.cascades) > 1)
        self.set_deterministic(os.environ.get("NGP_DETERMINISTIC", "0") == "1")

    def _init_exchange(self, model, dev, exchange):
        """world > 1: how the table's gradient and parameters travel (collectives, per-level-group overlap, direct peer memory)."""
        lvs = model.pos_encoder.levels_struct
        self._comm_stub = _exp.get("comm_stub", "0") == "1"
        # exchange="p2p" (round 6 prototype, ngp_hip/p2p.py): the table's reduce-scatter and all-gather as direct writes into the peers'
        # memory (hipIpc mappings) + flags instead of RCCL collectives -- one hop per phase on every xGMI link at once.  Needs the
        # sharded optimizer and the fp32 gradient; the 37.6 KB [MLP gradient | inf flag] all-reduce stays a collective.
        self._p2p = None
        if exchange not in ("rccl", "p2p"):
            raise ValueError("exchange must be 'rccl' or 'p2p'")
        if exchange == "p2p" and self.world > 1:
            if not self.shard or self.half or self._comm is not None:
                raise ValueError("exchange='p2p' needs the sharded optimizer, the fp32 encoder and fp32 gradient transport")
            from .p2p import PeerExchange
            stores = {"table": self.table_store}
            if self.copy16_store is not None:
                stores["copy16"] = self.copy16_store
            self._p2p = PeerExchange(self.rank, self.world, dev, self.shard_len, stores, group=self.group)
        self._pending_comm = []
        self._groups = None
        if (self.shard and _exp.get("comm_overlap", "0") == "1" and not self.half and self.hash_bwd == "sliced"
                and lvs.n_features == 2):
            self._groups = self._make_groups(lvs, _exp.get("comm_groups", "8,0"))

    def set_deterministic(self, on):
        if getattr(self, "_graph", None) is not None and bool(on) != self.deterministic:
            # the launch sequence (ray-ordered chain or one-launch march, which live list, which task plan) is baked into the captured
            # graphs: flipping the mode afterwards would silently keep replaying the old one
            raise RuntimeError("set_deterministic() after capture(): the captured graphs hold the previous mode's launches; "
                               "set the mode before capture()")
        self.deterministic = bool(on)
        self.model._ngp_deterministic = self.deterministic          # read by ngp_hip/occupancy.py
        return self

    def _scatter_mode(self):
        """The task plan this trainer's scatter-add launches run on, as NGP_BWD_PLAN_* bits.  The bits travel inside the level table
        handed to every ngp_hash_bwd_sliced_* call of the step (HashLevels.with_plan): the library holds no mode state."""
        return ((_lib_mod.BWD_PLAN_DETERMINISTIC if self.deterministic else 0)
                | (_lib_mod.BWD_PLAN_CONCENTRATED if self._concentrated else 0))

    def repack(self):
        """Rebuild the fp16 MFMA weight image from the fp32 master weights (call after loading a checkpoint into the
        model; the training step keeps it current by itself)."""
        ws = self.model._mlp_weights()
        self.sync_master()                    # sharded + 16-bit copy: never re-cast the copy from a master whose other shards are stale
        if self.table_bf16 is not None:
            self.model.pos_encoder._bf16_ver = None                             # force a re-cast of the bf16 table copy
            assert self.model.pos_encoder.table_bf16() is self.table_bf16
        if self.half:
            self.model.pos_encoder._f16_ver = None
            assert self.model.pos_encoder.table_f16().data_ptr() == self.table_f16.data_ptr()
        check(self.L.ngp_mlp_pack(*[_ptr(w) for w in ws], self.enc_pairs, _ptr(self.wpack), _stream()), "ngp_mlp_pack")

    # ------------------------------------------------------------------------------------------------ one step
    class _DevEvent:
        """Device-scope HIP event (ngp_event_*: no timing, no system-scope fence): orders two streams of this device."""

        def __init__(self, L):
            self.L = L
            h = ctypes.c_void_p()
            check(L.ngp_event_create(ctypes.byref(h)), "ngp_event_create")
            self.h = h

        def record(self, stream):
            check(self.L.ngp_event_record(self.h, ctypes.c_void_p(stream.cuda_stream)), "ngp_event_record")

        def wait(self, stream):
            check(self.L.ngp_stream_wait_event(ctypes.c_void_p(stream.cuda_stream), self.h), "ngp_stream_wait_event")

        def synchronize(self):
            check(self.L.ngp_event_synchronize(self.h), "ngp_event_synchronize")

        def __del__(self):
            try:
                self.L.ngp_event_destroy(self.h)
            except Exception:
                pass

    class _MarchSet:
        """Outputs of one batch's march.  Two sets alternate so the NEXT batch can be marched on a side stream while
        the current batch's encode / MLP / backward kernels (which read xyzs, dirs, deltas, ts) are still running."""

        def __init__(self, dev, n, max_samples):
            cap = n * max_samples
            f32 = dict(device=dev, dtype=torch.float32)
            self.n, self.cap = n, cap
            self.stage = torch.empty(cap, 2, **f32)
            self.counts = torch.empty(n, device=dev, dtype=torch.int32)
            self.rays_a = torch.empty(n, 3, device=dev, dtype=torch.int32)
            self.total = torch.zeros(1, device=dev, dtype=torch.int32)
            self.ctr = torch.zeros(2, device=dev, dtype=torch.int32)        # ngp_march_train_fused's counters (self-resetting)
            self.hits_t = torch.empty(n, 2, **f32)
            self.xyzs, self.dirs = torch.empty(cap, 3, **f32), torch.empty(cap, 3, **f32)
            self.deltas, self.ts = torch.empty(cap, **f32), torch.empty(cap, **f32)
            self.ready = None               # event recorded on the side stream when a prefetched march has finished
            self.issued_early = False       # ... and that march was issued at the START of the previous step (see step())
            self.src = None                 # the caller's (rays_o, rays_d) tensor OBJECTS + their versions the set was marched for
            self.held = None                # the (possibly converted) tensors the side-stream march reads: kept alive until reuse

        def marched_for(self, src):
            """True if this set holds a (possibly still running) prefetched march of exactly these caller tensors, unmodified
            since.  Identity + version, not data_ptr: a freed and reallocated tensor at the same address is a different batch."""
            k = self.src
            return (self.ready is not None and k is not None and src is not None and k[0] is src[0] and k[1] is src[1]
                    and k[2] == src[0]._version and k[3] == src[1]._version)


    @property
    def _live_total(self):
        """[1] int32 view: the live-sample count of the most recent step (diagnostics, bench.py)."""
        par = 1 - self._cur                                       # _launch flips _cur after choosing the step's set
        return self._live_pair[par:par + 1]

    def _march_sets(self, n):
        key = n
        sets = self._sets.get(key)
        if sets is None:
            sets = self._sets[key] = [self._MarchSet(self.dev, n, self.max_samples) for _ in range(2)]
            for m_ in sets:
                m_.ev_ready = self._DevEvent(self.L)           # recorded on the side stream behind a prefetched march
            sets[0].index, sets[1].index = 0, 1
        return sets

    def _coarse_bits(self, cfg, A):
        """The 8^3-block occupancy shortcut table, rebuilt only when the bitfield tensor was written to (its torch version counter
        moves on every in-place op, e.g. packbits / copy_; the raw-pointer kernels move it through ops._touched).  The validity key
        lives on the arena buffer, shared with the fused render() of the same (device, n_rays) (TrainArena.coarse_state)."""
        coarse, stale = A.coarse_state(cfg)
        if stale:
            check(self.L.ngp_bitfield_coarsen(_ptr(cfg.bitfield), cfg.cascades, cfg.grid_size, _ptr(coarse), _stream()),
                  "ngp_bitfield_coarsen")
        return coarse

    def _march(self, M, rays_o, rays_d, cfg, A, coarse=None, noise=None, shape=None):
        """ray-AABB + count/scan/write into march set M on the CURRENT stream."""
        L, st, n = self.L, _stream(), rays_o.shape[0]
        if coarse is None:
            coarse = self._coarse_bits(cfg, A)
        if shape is not None and noise is None and self.march_fused and self._march_rng and not self.deterministic:
            seed = int(torch.randint(0, 2**62, (), dtype=torch.int64))
            check(L.ngp_march_train_fused_shaped(_ptr(rays_o), _ptr(rays_d), _ptr(None), _ptr(cfg.bitfield), _ptr(coarse), _ptr(None), seed,
                                                 cfg.cascades, cfg.grid_size, cfg.scale, cfg.exp_step_factor, cfg.max_samples, n,
                                                 int(shape[0]), int(shape[1]), _ptr(M.stage), _ptr(M.ctr), _ptr(M.rays_a), _ptr(M.total),
                                                 _ptr(M.xyzs), _ptr(M.dirs), _ptr(M.deltas), _ptr(M.ts), st), "ngp_march_train_fused_shaped")
            return
        if noise is None and self.deterministic:
            # same counter-based jitter as the one-launch march draws in-kernel, as an explicit vector for the ray-order chain
            from .ops import rng_uniform
            noise = rng_uniform(int(torch.randint(0, 2**62, (), dtype=torch.int64)), n, self.dev)
        if noise is None and self.march_fused and self._march_rng:
            # the per-ray jitter (torch.rand_like, ray_march.py:138) is drawn inside the march kernel from a counter-based
            # generator keyed by (seed, ray); the seed comes from torch's CPU generator, so torch.manual_seed() still fixes the
            # jitter sequence -- and no uniform_ kernel sits on the (side) stream in front of the march
            seed = int(torch.randint(0, 2**62, (), dtype=torch.int64))
            check(L.ngp_march_train_fused_rng(_ptr(rays_o), _ptr(rays_d), _ptr(None), _ptr(cfg.bitfield), _ptr(coarse), seed,
                                              cfg.cascades, cfg.grid_size, cfg.scale, cfg.exp_step_factor, cfg.max_samples, n,
                                              _ptr(M.stage), _ptr(M.ctr), _ptr(M.rays_a), _ptr(M.total), _ptr(M.xyzs), _ptr(M.dirs),
                                              _ptr(M.deltas), _ptr(M.ts), st), "ngp_march_train_fused_rng")
            return
        if noise is None:
            noise = torch.rand(n, device=self.dev, dtype=torch.float32)                     # ray_march.py:138
        if self.march_fused and not self.deterministic:
            # one launch: count, block-wise allocation of the output ranges (rays in block-completion order, like the reference's
            # atomic packing), expansion.  NGP_EXPERIMENT march_fused=0: the count / scan / write chain (rays packed in ray order)
            check(L.ngp_march_train_fused(_ptr(rays_o), _ptr(rays_d), _ptr(None), _ptr(cfg.bitfield), _ptr(coarse), _ptr(noise),
                                          cfg.cascades, cfg.grid_size, cfg.scale, cfg.exp_step_factor, cfg.max_samples, n,
                                          _ptr(M.stage), _ptr(M.ctr), _ptr(M.rays_a), _ptr(M.total), _ptr(M.xyzs), _ptr(M.dirs),
                                          _ptr(M.deltas), _ptr(M.ts), st), "ngp_march_train_fused")
            return
        check(L.ngp_march_train_count_ex(_ptr(rays_o), _ptr(rays_d), _ptr(None), _ptr(cfg.bitfield), _ptr(coarse), _ptr(noise),
                                         cfg.cascades, cfg.grid_size, cfg.scale, cfg.exp_step_factor, cfg.max_samples, n,
                                         _ptr(M.stage), _ptr(M.counts), st), "ngp_march_train_count_ex")   # slab test inline
        check(L.ngp_march_train_scan(_ptr(M.counts), n, _ptr(M.rays_a), _ptr(M.total), st), "ngp_march_train_scan")
        check(L.ngp_march_train_write(_ptr(rays_o), _ptr(rays_d), _ptr(M.rays_a), _ptr(M.stage), cfg.max_samples, n,
                                      _ptr(M.xyzs), _ptr(M.dirs), _ptr(M.deltas), _ptr(M.ts), st), "ngp_march_train_write")

    def _launch(self, rays_o, rays_d, target, prefetch=None, src=None, src_next=None, noise=None):
        self.finish_comm()                   # (overlapped exchange: the previous step's all-gathers, before the table is read)
        n = rays_o.shape[0]
        cfg = RenderConfig.cached(self.model, self.exp_step_factor, self.T_threshold, self.max_samples)
        A = TrainArena.get(self.dev, n, self.max_samples)
        sets = self._march_sets(n)
        M = sets[self._cur]
        hit = noise is None and M.marched_for(src)               # an explicit jitter vector always re-marches
        self.prefetch_hits += int(hit)
        if M.ready is not None:
            # whatever the side stream did to this set has to be finished before the main stream reads OR rewrites it (a stale
            # prefetch for other rays would otherwise race with the re-march below on M.stage / counts / xyzs).
            # Round 6: a stream-side wait for an event of ANOTHER queue is a barrier packet that costs the main stream ~10 us even when
            # the event completed long ago (timing run without it: 0.4745 -> 0.4635 ms per step).  NGP_EXPERIMENT prefetch_host_wait=1 lets
            # the HOST wait instead where the march was issued at the START of the previous step (done ~100 us into that step): no packet,
            # the host at most one step ahead.  Measured -5 us in three alternating pairs on one box, nothing (0.461 against 0.462 ms) in
            # pairs on three other boxes, and one unexplained cluster of 0.51 ms runs on three boxes with it on: it couples the device to
            # the host's pace for a gain inside the box-to-box spread, so it stays OFF by default (profiles/r06_scatter_add_valu_experiment.txt (15)).
            if self._host_wait_ok and M.issued_early and self._graph is None:
                M.ready.synchronize()
            else:
                M.ready.wait(torch.cuda.current_stream())
        if not hit:
            self._march(M, rays_o, rays_d, cfg, A, noise=noise)
        M.ready, M.src, M.held = None, None, None
        hook = None
        if prefetch is not None and (prefetch[0].shape != rays_o.shape or prefetch[1].shape != rays_d.shape):
            raise ValueError("prefetch rays must have the shape of the current batch (the two march buffers are sized per batch size)")
        if prefetch is not None:
            # software pipelining across steps: the march only depends on the rays and the occupancy bitfield, never on
            # the weights, and it is latency-bound (few resident waves) -- run the NEXT batch's march on a side stream
            # underneath this step's kernels.
            nxt = sets[1 - self._cur]

            at, shape, side = self._prefetch_at, self._march_shape, self._side
            if self._adaptive_prefetch:
                marched = ctypes.c_int32.from_address(self._marched_host.value).value
                if marched <= self._MARCH_NARROW_MAX:
                    at, shape, side = 0, (4, 0), self._side_default
                else:
                    at, shape, side = 3, (4, 0), self._side_low
            self._hook_at = at

            def hook():
                start = self._ev_start
                start.record(torch.cuda.current_stream())                   # everything that still reads `nxt` is before this
                with torch.cuda.stream(side):
                    start.wait(side)
                    if nxt.ready is not None:
                        nxt.ready.wait(side)                                # an unconsumed earlier prefetch into the same set
                    self._march(nxt, prefetch[0], prefetch[1], cfg, A, shape=shape)
                    nxt.ready = nxt.ev_ready
                    nxt.ready.record(side)
                    nxt.issued_early = at == 0
                    if self._marched_host is not None:                    # behind the march, on the side stream: nobody waits
                        check(self.L.ngp_copy_to_host_async(self._marched_host, _ptr(nxt.total), 4, ctypes.c_void_p(side.cuda_stream)),
                              "ngp_copy_to_host_async")
                nxt.src = None if src_next is None else (src_next[0], src_next[1], src_next[0]._version, src_next[1]._version)
                nxt.held = prefetch          # (possibly temporaries of step()): alive until the set is consumed or re-marched
            if at == 0 or self._graph is not None:
                hook(); hook = None
        # The fused live list relies on the counter of this step's parity being 0 at launch (the composite kernel of the PREVIOUS
        # step cleared it).  A step that raised after the flip below never ran that kernel: if the last launch did not complete,
        # clear both counters on the stream before going on (stale counts would be added to, silently).
        if self._launch_incomplete:
            self._live_pair.zero_()
            if self._chunk_counts is not None:
                self._chunk_counts.zero_()
        self._launch_incomplete = True
        cur, self._cur = self._cur, 1 - self._cur
        if self._graph is None:
            stats = self._shade(M, n, target, cfg, A, hook)
            self._launch_incomplete = False
            return stats
        # hipGraph mode: the shading / backward / optimizer chain of march set `cur` is one graph launch
        if n != self._graph_n:
            raise ValueError("graph mode was captured for %d rays per step" % self._graph_n)
        check(self.L.ngp_stage_batch(_ptr(target), _ptr(self._static_target), _ptr(None), _ptr(None), _ptr(None), _ptr(None), 3 * n,
                                     _stream()), "ngp_stage_batch")
        if cur not in self._graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                stats = self._shade(M, n, self._static_target, cfg, A)
            self._graph[cur] = (g, stats)
        g, stats = self._graph[cur]
        g.replay()
        self._launch_incomplete = False
        return stats

    def _shade(self, M, n, target, cfg, A, hook=None):
        """Everything after the march: encode, MLPs, composite + loss, backward, [all-reduce], optimizer -- on the current stream."""
        L, st, dev = self.L, _stream(), self.dev
        i32 = dict(device=dev, dtype=torch.int32)
        f32 = dict(device=dev, dtype=torch.float32)
        rays_a, total = M.rays_a, M.total
        vr_per_ray = torch.empty(n, **i32)
        opacity, depth, rgb = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, 3, **f32)
        sf, si = self.state_f, self.state_i
        sq_err = torch.empty(n, **f32)
        found = ctypes.c_void_p(si.data_ptr() + 4 * _SI_FOUND_INF)
        P = self.enc_pairs
        done = False
        if self.chunked:
            done, hook = self._shade_chunked(M, n, cfg, A, P, st, hook)
        if not done:
            hook = self._shade_all(M, cfg, A, P, st, total, hook)
        par = M.index
        live_total, live_next = self._live_pair[par:par + 1], self._live_pair[1 - par:2 - par]
        fused_live = False
        if self.distortion_loss_w > 0:
            sq_err = self._composite_with_distortion(A, M, target, cfg, n, vr_per_ray, opacity, depth, rgb)
        else:
            # composite forward + MSE gradient + composite backward, one launch -- and, as a by-product, the compacted list of
            # the LIVE samples (those in front of each ray's early-termination point; the rest have exact-zero gradients) that
            # the MLP backward and the scatter-add run over
            fused_live = self.live_backward and not self.deterministic      # (deterministic: the ray-ordered list of ngp_live_compact)
            check(L.ngp_composite_train_fused_live(_ptr(A.sigmas), _ptr(A.rgbs), 1, _ptr(M.deltas), _ptr(M.ts), _ptr(rays_a),
                                                   _ptr(target), self.bg, _ptr(sf), cfg.T_threshold, n, _ptr(vr_per_ray), _ptr(opacity),
                                                   _ptr(depth), _ptr(rgb), _ptr(A.ws), _ptr(A.d_sigmas), _ptr(A.d_rgbs), _ptr(sq_err),
                                                   _ptr(A.live_idx if fused_live else None), _ptr(live_total if fused_live else None),
                                                   _ptr(live_next if fused_live else None), st), "ngp_composite_train_fused_live")
        live_idx = A.live_idx
        if self.live_backward and not fused_live:                 # (distortion-loss path: its composite is the operator chain)
            if self.deterministic:                                # the ray-ORDERED list (every block scans all rays: 152 us at 65 536 rays)
                check(L.ngp_live_compact(_ptr(rays_a), _ptr(vr_per_ray), n, _ptr(A.live_off(n)), _ptr(live_idx), _ptr(live_total), st),
                      "ngp_live_compact")
                live_next.zero_()
            else:                                                 # block-completion order, one atomic per 64 rays; clears the other counter
                check(L.ngp_live_list(_ptr(rays_a), _ptr(vr_per_ray), n, _ptr(live_idx), _ptr(live_total), _ptr(live_next), st),
                      "ngp_live_list")
        elif not fused_live:
            live_next.zero_()
        if self.live_backward:
            cnt = live_total
        else:
            live_idx, cnt = None, total
        # the scatter-add's prepass (hit bitmaps + compact positions; needs the positions and the live list only).  It used to run on
        # a second stream underneath the MLP backward, but the two share the VALU (87 us overlapped vs 43 alone) and the
        # cross-stream join cost 23 us between the prepass's end and the scatter-add's start: in line it is 17 us per step faster
        sliced = self.hash_bwd == "sliced"          # (half2 encoder: same prepass, main pass with its fp16 arithmetic + fp16 table)
        det = self._scatter_mode()
        lvp = self._lvp = cfg.levels.with_plan(det)           # the level table + this step's plan bits: prep, main and adam_prefix see the same
        if hook is not None and self._hook_at == 2.5:
            hook(); hook = None                                             # position 2.5: under the prepass, the MLP backward and the scatter-add
        if sliced:
            ws = A.sliced_ws(cfg.levels)
            rc = L.ngp_hash_bwd_sliced_prep(_ptr(M.xyzs), ctypes.byref(lvp), A.cap, _ptr(cnt), _ptr(live_idx), 1, cfg.lo, cfg.hi,
                                            _ptr(ws), ws.numel(), st)
            if rc == -2:
                # level table not expressible as <= 64 LDS slices per level: the float-atomic kernel from here on.  The overlapped
                # exchange is built on per-level-group launches of the sliced kernel and shards the optimizer BY GROUP; falling
                # through to the contiguous shard layout with the group layout still recorded would make state_dict() /
                # sync_master() gather the wrong ranges -- a trainer configured that way cannot continue
                if self._groups is not None:
                    raise RuntimeError("NGP_EXPERIMENT comm_overlap=1 needs the LDS-sliced scatter-add, which this level table does not fit "
                                       "(ngp_hash_bwd_sliced_prep returned -2); unset NGP_EXPERIMENT comm_overlap")
                self.hash_bwd, sliced = "atomic", False
            else:
                check(rc, "ngp_hash_bwd_sliced_prep")
        # weight gradients leave the launch as per-block slabs (plain stores) instead of 256 x 9408 same-address float atomics (13 us
        # of the launch); the slabs are added up by the prologue launch below, or -- when the gradient is needed before that (an
        # exchange between ranks, compute_gradients) -- by a launch of its own right here
        if hook is not None and self._hook_at == 2.75:
            hook(); hook = None                                             # position 2.75: under the MLP backward and the scatter-add
        if self._dw_atomic:                                   # NGP_EXPERIMENT mlp_dw=atomic: round 3's flush, for A/B runs
            check(L.ngp_mlp_bwd_live(_ptr(A.enc), _ptr(M.dirs), _ptr(self.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap, _ptr(cnt),
                                     _ptr(live_idx), P, _ptr(A.d_enc), _ptr(self.mlp_grad), found, st), "ngp_mlp_bwd_live")
            n_parts = 0
        else:
            n_parts = L.ngp_mlp_bwd_live_parts(_ptr(A.enc), _ptr(M.dirs), _ptr(self.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap,
                                               _ptr(cnt), _ptr(live_idx), P, _ptr(A.d_enc), _ptr(self.mlp_parts), found, st)
            if n_parts <= 0:
                check(n_parts or -1, "ngp_mlp_bwd_live_parts")
        # ... summed by the head of the scatter-add launch (its first 147 persistent workgroups, ~2 us hidden in 200) on one GPU with
        # the LDS-sliced scatter-add; by the prologue launch when the scatter-add is another kernel; by a launch of its own when the
        # gradient is needed earlier (an exchange between ranks, compute_gradients)
        single = self.world == 1 and not self._grads_only and n_parts > 0 and self._groups is None   # (overlapped tail: no slab hand-over)
        reduce_in_scatter = single and sliced and not self.half and _exp.get("mlp_dw_reduce", "scatter") == "scatter"
        reduce_in_prologue = single and not reduce_in_scatter
        if n_parts > 0 and not single:
            check(L.ngp_mlp_dw_reduce(_ptr(self.mlp_parts), n_parts, _ptr(self.mlp_grad), st), "ngp_mlp_dw_reduce")
        if hook is not None and self._hook_at <= 3:
            hook(); hook = None                                             # position 3: under the scatter-add and the optimizer
        if reduce_in_scatter and self._flush_adam:
            npre = self._adam_prefix.get(det)
            if npre is None:
                npre = int(L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lvp)))
                npre = self._adam_prefix[det] = npre if npre % 4 == 0 else -2
            if npre >= 0:
                return self._tail_flush_adam(A, M, cfg, cnt, P, ws, n_parts, st, hook, total, vr_per_ray, rgb, opacity, depth, sq_err, npre)
        if self._groups is not None and sliced and not self._grads_only:
            self._tail_overlapped(A, cfg, cnt, P, ws, found, st, hook, None)
            return {"rm_samples": total, "vr_per_ray": vr_per_ray, "rgb": rgb, "opacity": opacity, "depth": depth, "rays_a": rays_a,
                    "deltas": M.deltas, "ts": M.ts, "sq_err": sq_err}
        if reduce_in_scatter:
            check(L.ngp_hash_bwd_sliced_main_slabs(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad),
                                                   0, found, _ptr(ws), ws.numel(), _ptr(self.mlp_parts), n_parts,
                                                   _ptr(self.mlp_grad), st), "ngp_hash_bwd_sliced_main_slabs")
        elif self.half and sliced:
            check(L.ngp_hash_bwd_sliced_main_f16(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad), found,
                                                 _ptr(ws), ws.numel(), st), "ngp_hash_bwd_sliced_main_f16")
        elif self.half:
            check(L.ngp_hash_bwd_f16_live(_ptr(M.xyzs), _ptr(A.d_enc), ctypes.byref(cfg.levels), A.cap, _ptr(cnt), _ptr(live_idx), 1,
                                          cfg.lo, cfg.hi, P, _ptr(self.table_grad), found, st), "ngp_hash_bwd_f16_live")
        elif sliced:
            check(L.ngp_hash_bwd_sliced_main(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad), found,
                                             _ptr(ws), ws.numel(), st), "ngp_hash_bwd_sliced_main")
        else:
            check(L.ngp_hash_bwd_f32_live(_ptr(M.xyzs), _ptr(A.d_enc), ctypes.byref(cfg.levels), A.cap, _ptr(cnt), _ptr(live_idx), 1,
                                          cfg.lo, cfg.hi, P, _ptr(self.table_grad), found, st), "ngp_hash_bwd_f32_live")
        if hook is not None:
            hook(); hook = None                                             # position 4: under the gradient exchange + optimizer
        sharded = self.shard and not self._grads_only                      # (gradient diagnostics use the plain all-reduce)
        if sharded:
            self._exchange_sharded(found, st)
        else:
            if self.world > 1:
                self._all_reduce()
            if self.half:   # f16 sums overflow easily: GradScaler's check must see the ACCUMULATED (and reduced) gradient
                check(L.ngp_check_finite_f16(_ptr(self.table_grad), self.table_grad.numel(), found, st), "ngp_check_finite_f16")
        if self._grads_only:
            return {"rm_samples": total, "vr_per_ray": vr_per_ray, "rgb": rgb, "opacity": opacity, "depth": depth, "rays_a": rays_a,
                    "deltas": M.deltas, "ts": M.ts, "sq_err": sq_err}
        if reduce_in_prologue:
            check(L.ngp_train_prologue_reduce(_ptr(sf), _ptr(si), self.lr0, self.eta_min, self.t_max, self.beta1, self.beta2,
                                              self.growth, self.backoff, self.growth_interval, _ptr(self.mlp_parts), n_parts,
                                              _ptr(self.mlp_grad), st), "ngp_train_prologue_reduce")
        else:
            check(L.ngp_train_prologue(_ptr(sf), _ptr(si), self.lr0, self.eta_min, self.t_max, self.beta1,
                                       self.beta2, self.growth, self.backoff, self.growth_interval, st), "ngp_train_prologue")
        # Adam on the table (+ its bf16 copy) and on the MLP weights + the fp16 fragment repack the next step needs: one launch
        kind = 2 if self.half else (1 if self.table_bf16 is not None else 0)
        if sharded:
            # Adam on THIS rank's 1/world of the table (its gradient shard is already the cross-rank average); the MLP block of the
            # launch runs replicated on the all-reduced MLP gradient; then the updated parameters travel back
            lo = self.rank * self.shard_len
            sl = slice(lo, lo + self.shard_len)
            c16 = self.copy16_store[sl] if self.copy16_store is not None else None
            check(L.ngp_adam_all_ex(_ptr(self.table_store[sl]), _ptr(self.shard_grad), int(self.half), _ptr(self.table_m[sl]),
                                    _ptr(self.table_v[sl]), self.shard_len, _ptr(c16), kind, _ptr(self.mlp_flat), _ptr(self.mlp_grad),
                                    _ptr(self.mlp_m), _ptr(self.mlp_v), _ptr(sf), _ptr(si), self.beta1, self.beta2, self.eps, P,
                                    _ptr(self.wpack), st), "ngp_adam_all_ex")
            if self.copy16_store is not None:
                self._all_gather(self.copy16_store, sl)          # the forward (and the occupancy update) read the 16-bit copy only
                self._master_stale = True                        # fp32 master of the other ranks' shards: sync_master() on demand
            else:
                self._all_gather(self.table_store, sl)
        else:
            copy16 = self.copy16_store[:self.nt] if self.copy16_store is not None else None
            check(L.ngp_adam_all_ex(_ptr(self.table), _ptr(self.table_grad), int(self.half), _ptr(self.table_m), _ptr(self.table_v),
                                    self.table.numel(), _ptr(copy16), kind, _ptr(self.mlp_flat), _ptr(self.mlp_grad), _ptr(self.mlp_m),
                                    _ptr(self.mlp_v), _ptr(sf), _ptr(si), self.beta1, self.beta2, self.eps, P, _ptr(self.wpack), st),
                  "ngp_adam_all_ex")
        return {"rm_samples": total, "vr_per_ray": vr_per_ray, "rgb": rgb, "opacity": opacity, "depth": depth, "rays_a": rays_a,
                "deltas": M.deltas, "ts": M.ts, "sq_err": sq_err}

    def _shade_all(self, M, cfg, A, P, st, total, hook):
        """Encode + MLP forward over every marched sample (reference networks.py:136-166 on all of raymarching_train's output)."""
        L = self.L
        if self.half:
            check(L.ngp_hash_fwd_f16_ex(_ptr(M.xyzs), _ptr(self.table_f16), ctypes.byref(cfg.levels), A.cap, _ptr(total), 1, cfg.lo,
                                        cfg.hi, P, _ptr(A.enc), st), "ngp_hash_fwd_f16_ex")
        elif self.table_bf16 is not None:
            check(L.ngp_hash_fwd_bf16_ex(_ptr(M.xyzs), _ptr(self.table_bf16), ctypes.byref(cfg.levels), A.cap, _ptr(total), 1, cfg.lo,
                                         cfg.hi, P, _ptr(A.enc), st), "ngp_hash_fwd_bf16_ex")
        else:
            check(L.ngp_hash_fwd_f32_ex(_ptr(M.xyzs), _ptr(self.table), ctypes.byref(cfg.levels), A.cap, _ptr(total), 1, cfg.lo,
                                        cfg.hi, P, _ptr(A.enc), st), "ngp_hash_fwd_f32_ex")
        if hook is not None and self._hook_at == 1:
            hook(); hook = None
        check(L.ngp_mlp_fwd_ex(_ptr(A.enc), _ptr(M.dirs), _ptr(self.wpack), A.cap, _ptr(total), P, _ptr(A.sigmas), _ptr(A.rgbs), st),
              "ngp_mlp_fwd_ex")
        if hook is not None and self._hook_at == 2:
            hook(); hook = None
        return hook

    def _shade_chunked(self, M, n, cfg, A, P, st, hook):
        """Encode + MLP forward in rounds over the samples compositing can still reach (see __init__; csrc/composite.hip
        chunk_schedule_kernel).  Returns False when the level table does not fit the list encoder (the caller shades everything)."""
        L = self.L
        R = len(self._chunk_rounds)
        T_state = self._chunk_state(n)
        par = M.index
        c_cur = self._chunk_counts.data_ptr() + 4 * R * par             # this step's list lengths; the other parity's set is cleared
        c_oth = self._chunk_counts.data_ptr() + 4 * R * (1 - par)       # round by round for the next step
        table, kind = (self.table_bf16, 1) if self.table_bf16 is not None else (self.table, 0)
        lst = A.live_idx                    # (free until the composite kernel writes the backward's live list into it)
        for r, (b, l, pb) in enumerate(self._chunk_rounds):
            cnt = ctypes.c_void_p(c_cur + 4 * r)
            check(L.ngp_chunk_schedule(_ptr(M.rays_a), _ptr(A.sigmas), _ptr(M.deltas), n, b, l, pb, 0.5 * cfg.T_threshold, _ptr(T_state),
                                       _ptr(lst), cnt, ctypes.c_void_p(c_oth + 4 * r), st), "ngp_chunk_schedule")
            rc = L.ngp_hash_fwd_list(_ptr(M.xyzs), _ptr(table), kind, ctypes.byref(cfg.levels), A.cap, cnt, _ptr(lst), 1,
                                     cfg.lo, cfg.hi, P, _ptr(A.enc), st)
            if rc == -2 and r == 0:                                        # level table outside the list encoder: shade everything
                self._chunk_counts.zero_()
                self.chunked = False
                return False, hook
            check(rc, "ngp_hash_fwd_list")
            if r == 0 and hook is not None and self._hook_at == 1:
                hook(); hook = None
            check(L.ngp_mlp_fwd_list(_ptr(A.enc), _ptr(M.dirs), _ptr(self.wpack), A.cap, cnt, _ptr(lst), P, _ptr(A.sigmas),
                                     _ptr(A.rgbs), st), "ngp_mlp_fwd_list")
        if hook is not None and self._hook_at == 2:
            hook(); hook = None
        return True, hook

    def _chunk_state(self, n):
        """Buffers of the chunked forward for n rays per step: [2, rounds] list counters (one set per step parity, all zero between
        steps) and the per-ray transmittance carried from round to round.  Allocated on first use -- by capture() in graph mode."""
        if self._chunk_counts is None:
            self._chunk_counts = torch.zeros(2, len(self._chunk_rounds), device=self.dev, dtype=torch.int32)
        T_state = self._chunk_T.get(n)
        if T_state is None:
            T_state = self._chunk_T[n] = torch.empty(n, device=self.dev, dtype=torch.float32)
        return T_state

    def shaded_samples(self):
        """Samples the most recent step shaded (chunked forward; None when every marched sample is shaded).  One host read."""
        if not self.chunked or self._chunk_counts is None:
            return None
        return int(self._chunk_counts[1 - self._cur].sum())

    def _tail_flush_adam(self, A, M, cfg, cnt, P, ws, n_parts, st, hook, total, vr_per_ray, rgb, opacity, depth, sq_err, npre):
        """Prologue -> scatter-add with the optimizer in its flush -> Adam on the replicated coarse levels + the MLP (one GPU, fp32
        master table with or without the bf16 copy; see __init__)."""
        L, sf, si, lvp = self.L, self.state_f, self.state_i, self._lvp
        copy16 = self.copy16_store[:self.nt] if self.copy16_store is not None else None
        if self._fold_prologue:
            # the GradScaler / schedule decision is evaluated inside the scatter-add launch (no one-thread launch in front of it)
            check(L.ngp_hash_bwd_sliced_main_adam_step(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad),
                                                       _ptr(ws), ws.numel(), _ptr(self.mlp_parts), n_parts, _ptr(self.mlp_grad),
                                                       _ptr(self.table), _ptr(self.table_m), _ptr(self.table_v), _ptr(copy16), _ptr(sf),
                                                       _ptr(si), self.lr0, self.eta_min, self.t_max, self.beta1, self.beta2, self.eps,
                                                       self.growth, self.backoff, self.growth_interval, st),
                  "ngp_hash_bwd_sliced_main_adam_step")
        else:
            check(L.ngp_train_prologue(_ptr(sf), _ptr(si), self.lr0, self.eta_min, self.t_max, self.beta1, self.beta2, self.growth,
                                       self.backoff, self.growth_interval, st), "ngp_train_prologue")
            check(L.ngp_hash_bwd_sliced_main_adam(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad), _ptr(ws),
                                                  ws.numel(), _ptr(self.mlp_parts), n_parts, _ptr(self.mlp_grad), _ptr(self.table),
                                                  _ptr(self.table_m), _ptr(self.table_v), _ptr(copy16), _ptr(sf), _ptr(si), self.beta1,
                                                  self.beta2, self.eps, st), "ngp_hash_bwd_sliced_main_adam")
        if hook is not None:
            hook()                                                          # position 4
        kind = 1 if copy16 is not None else 0
        if npre > 0:
            check(L.ngp_adam_all_ex(_ptr(self.table), _ptr(self.table_grad), 0, _ptr(self.table_m), _ptr(self.table_v), npre,
                                    _ptr(copy16), kind, _ptr(self.mlp_flat), _ptr(self.mlp_grad), _ptr(self.mlp_m), _ptr(self.mlp_v),
                                    _ptr(sf), _ptr(si), self.beta1, self.beta2, self.eps, P, _ptr(self.wpack), st), "ngp_adam_all_ex")
        else:
            check(L.ngp_adam_mlp_pack(_ptr(self.mlp_flat), _ptr(self.mlp_grad), _ptr(self.mlp_m), _ptr(self.mlp_v), _ptr(sf), _ptr(si),
                                      self.beta1, self.beta2, self.eps, P, _ptr(self.wpack), st), "ngp_adam_mlp_pack")
        return {"rm_samples": total, "vr_per_ray": vr_per_ray, "rgb": rgb, "opacity": opacity, "depth": depth, "rays_a": M.rays_a,
                "deltas": M.deltas, "ts": M.ts, "sq_err": sq_err}

    def _composite_with_distortion(self, A, M, target, cfg, n, vr_per_ray, opacity, depth, rgb):
        """loss = MSE + w * mean(distortion) (train.py:193-195): the distortion gradient w.r.t. the sample weights has to
        exist before the composite backward runs, so the single fused launch becomes composite fwd -> distortion fwd/bwd ->
        MSE gradient -> composite bwd (all the same kernels the operator path uses).  Returns the per-ray squared error
        stand-in used for logging (MSE part only)."""
        L, st, sf = self.L, _stream(), self.state_f
        f32 = dict(device=self.dev, dtype=torch.float32)
        rays_a = M.rays_a
        check(L.ngp_composite_train_fwd(_ptr(A.sigmas), _ptr(A.rgbs), 1, _ptr(M.deltas), _ptr(M.ts), _ptr(rays_a), cfg.T_threshold,
                                        n, _ptr(vr_per_ray), _ptr(opacity), _ptr(depth), _ptr(rgb), _ptr(A.ws), st),
              "ngp_composite_train_fwd")
        dist_loss = torch.zeros(n, **f32)
        ws_inc, wts_inc, g_ws = A.scratch("ws_inc"), A.scratch("wts_inc"), A.scratch("g_ws")
        check(L.ngp_distortion_fwd(_ptr(A.ws), _ptr(M.deltas), _ptr(M.ts), _ptr(rays_a), n, _ptr(dist_loss), _ptr(ws_inc),
                                   _ptr(wts_inc), st), "ngp_distortion_fwd")
        # d(w * mean(dist)) / d dist[r] = w / n, loss-scaled on the device like every other gradient
        g_dist = (sf[_SF_LOSS_SCALE] * (self.distortion_loss_w / n)).expand(n).contiguous()
        check(L.ngp_distortion_bwd(_ptr(g_dist), _ptr(A.ws), _ptr(M.deltas), _ptr(M.ts), _ptr(ws_inc), _ptr(wts_inc), _ptr(rays_a), n,
                                   _ptr(g_ws), st), "ngp_distortion_bwd")
        g_rgb, g_op, sq_err = torch.empty(n, 3, **f32), torch.empty(n, **f32), torch.empty(n, **f32)
        # per ray, from as many blocks as the batch needs; the squared error leaves per ray (summed by whoever logs the loss)
        check(L.ngp_mse_loss_grad_rays(_ptr(rgb), _ptr(opacity), _ptr(target), self.bg, n, _ptr(sf), _ptr(g_rgb), _ptr(g_op), _ptr(sq_err), st),
              "ngp_mse_loss_grad_rays")
        check(L.ngp_composite_train_bwd(_ptr(g_op), _ptr(None), _ptr(g_rgb), _ptr(g_ws), _ptr(A.sigmas), _ptr(A.rgbs), 1, _ptr(M.deltas),
                                        _ptr(M.ts), _ptr(rays_a), _ptr(opacity), _ptr(depth), _ptr(rgb), _ptr(A.ws), cfg.T_threshold, n,
                                        _ptr(A.d_sigmas), _ptr(A.d_rgbs), st), "ngp_composite_train_bwd")
        self._dist_loss = dist_loss
        return sq_err

    # ---- world > 1, sharded optimizer: reduce-scatter -> Adam on the own shard -> all-gather ------------------------------
    def _nccl(self):
        return dist.get_backend(self.group) == "nccl"

    # ---- overlapped exchange: one launch + one reduce-scatter / Adam / all-gather per level group (ngp_hip/dist.py) ------------
    def _make_groups(self, lvs, spec):
        from .dist import GroupedExchange, make_level_groups
        groups = make_level_groups([int(lvs.offset[l]) for l in range(int(lvs.n_levels))], int(lvs.n_features), self.nt_pad,
                                   self.world, self.rank, spec)
        if groups[0].hi <= groups[0].lo:
            raise ValueError("the first level group leaves rank %d without a chunk" % self.rank)
        self._gx = GroupedExchange(groups, self.rank, self.world, self.dev, group=self.group,
                                   comm_dtype=self._comm.dtype if self._comm is not None else None, timed=self._timed)
        return groups

    def _rs_async(self, g):
        self._gx.stub = self._comm_stub
        return self._gx.rs_async(g, self.table_grad_store)

    def _ag_async(self, g, store):
        self._gx.stub = self._comm_stub
        return self._gx.ag_async(g, store)

    def _timed_wait(self, name, fn):
        """Overlapped exchange: what the step's stream WAITS for a collective that was issued earlier (sampled steps only)."""
        return self._timed(name, fn)

    def finish_comm(self):
        """Overlapped exchange: make the step's stream wait for the all-gathers of the previous step (before anything reads the
        table: the next forward, an occupancy update, a checkpoint)."""
        pend, self._pending_comm = self._pending_comm, []
        for fin in pend:
            fin()

    def _gather_groups(self, store):
        """state_dict() / sync_master() of the overlapped layout: all-gather a full-size buffer group by group (synchronous)."""
        for g in self._groups:
            self._ag_async(g, store)()

    def _tail_overlapped(self, A, cfg, cnt, P, ws, found, st, hook, reduce_parts):
        """Scatter-add, gradient exchange and optimizer of one step with the exchange overlapped (see __init__)."""
        L, sf, si, lvp = self.L, self.state_f, self.state_i, self._lvp
        max_blocks = int(_exp.get("comm_scatter_blocks", "240"))     # leave CUs for RCCL's workgroups beside the later launches
        fins = []
        for k, g in enumerate(self._groups):
            check(L.ngp_hash_bwd_sliced_main_levels(_ptr(A.d_enc), ctypes.byref(lvp), A.cap, _ptr(cnt), P, _ptr(self.table_grad),
                                                    found, _ptr(ws), ws.numel(), g.mask, 0 if k == 0 else max_blocks, st),
                  "ngp_hash_bwd_sliced_main_levels")
            fins.append(self._rs_async(g))
        if hook is not None:
            hook()                                                          # the next batch's march: under the exchange
        # [MLP gradient | inf flag]: every rank has to take the same skip / step decision; behind the last reduce-scatter
        flag_i = si[_SI_FOUND_INF:_SI_FOUND_INF + 1]
        self._flag_f.copy_(flag_i)
        small_work = None
        if not self._comm_stub:
            if self._nccl():
                small_work = dist.all_reduce(self.small_bucket, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                small_work = dist.all_reduce(self.small_bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        def wait_small():
            if small_work is not None:
                small_work.wait()
        self._timed_wait("wait_all_reduce_mlp_grad_and_flag", wait_small)
        if small_work is not None and not self._nccl():
            self.small_bucket.div_(self.world)
        flag_i.copy_(self._flag_f != 0)
        self._flag_f.zero_()
        check(L.ngp_train_prologue(_ptr(sf), _ptr(si), self.lr0, self.eta_min, self.t_max, self.beta1, self.beta2, self.growth,
                                   self.backoff, self.growth_interval, st), "ngp_train_prologue")
        kind = 1 if self.table_bf16 is not None else 0
        back = self.copy16_store if self.copy16_store is not None else self.table_store
        for k, (g, fin) in enumerate(zip(self._groups, fins)):
            fin()                                                           # this group's averaged gradient chunk has arrived
            n_own = g.hi - g.lo
            if n_own > 0:
                sl = slice(g.lo, g.hi)
                c16 = self.copy16_store[sl] if self.copy16_store is not None else None
                mlp = (self.mlp_flat, self.mlp_grad, self.mlp_m, self.mlp_v) if k == 0 else (None, None, None, None)
                check(L.ngp_adam_all_ex(_ptr(self.table_store[sl]), _ptr(g.shard), 0, _ptr(self.table_m[sl]), _ptr(self.table_v[sl]),
                                        n_own, _ptr(c16), kind, _ptr(mlp[0]), _ptr(mlp[1]), _ptr(mlp[2]), _ptr(mlp[3]), _ptr(sf),
                                        _ptr(si), self.beta1, self.beta2, self.eps, P, _ptr(self.wpack), st), "ngp_adam_all_ex")
            self._pending_comm.append(self._ag_async(g, back))
        if self.copy16_store is not None:
            self._master_stale = True

    def _exchange_sharded(self, found, st):
        """(1) reduce-scatter of the padded table gradient: this rank receives the cross-rank AVERAGE of shard `rank` (MSE is a
        mean over the local ray shard); the local accumulator is cleared for the next step.  (2) one small all-reduce of
        [MLP gradient | inf flag] (37.6 KB): the MLP Adam runs replicated, and every rank has to take the same skip / step
        decision (the half2 encoder's f16 gradient shard is scanned for inf first: f16 sums overflow)."""
        src = self.table_grad_store
        if self._comm is not None:                                 # 16-bit transport of the fp32 gradient (bench.py --comm bf16)
            self._comm.copy_(src)
            self._reduce_scatter(self._comm_shard, self._comm)
            self.shard_grad.copy_(self._comm_shard)
        else:
            self._reduce_scatter(self.shard_grad, src)
        src.zero_()
        if self.half:
            check(self.L.ngp_check_finite_f16(_ptr(self.shard_grad), self.shard_grad.numel(), found, st), "ngp_check_finite_f16")
        flag_i = self.state_i[_SI_FOUND_INF:_SI_FOUND_INF + 1]
        self._flag_f.copy_(flag_i)
        def small():
            if self._comm_stub:
                return
            if self._nccl():
                dist.all_reduce(self.small_bucket, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.small_bucket, op=dist.ReduceOp.SUM, group=self.group)
                self.small_bucket.div_(self.world)
        self._timed("all_reduce_mlp_grad_and_flag", small)
        flag_i.copy_(self._flag_f != 0)
        self._flag_f.zero_()

    def _timed(self, name, fn):
        """Run one collective; with a bench probe attached, bracket it with HIP events on the current stream (a synchronous
        torch.distributed collective makes this stream wait for the RCCL stream, so the bracket contains the transfer)."""
        probe = self.comm_probe() if self.comm_probe is not None else None
        if not probe or not probe[0]:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        probe[1].append((name, e0, e1))
        return out

    def comm_bytes_per_step(self):
        """Payload bytes this rank hands to the collectives per step (not the wire bytes: a ring moves (N-1)/N of it per phase)."""
        if self.world <= 1:
            return {}
        small = self.small_bucket.numel() * self.small_bucket.element_size()
        if self.shard:
            g = self._comm if self._comm is not None else self.table_grad_store
            back = self.copy16_store if self.copy16_store is not None else self.table_store
            return {"reduce_scatter_table_grad": g.numel() * g.element_size(), "all_reduce_mlp_grad_and_flag": small,
                    "all_gather_table": back.numel() * back.element_size()}
        buf = self._comm if self._comm is not None else self.grad_flat
        out = {"all_reduce_flat_bucket": buf.numel() * buf.element_size()}
        if self.half:
            out["all_reduce_f16_table_grad"] = self.table_grad.numel() * 2
        return out

    def _reduce_scatter(self, out, inp):
        from .dist import reduce_scatter_avg
        if self._comm_stub:
            return out.copy_(inp[self.rank * out.numel():(self.rank + 1) * out.numel()])
        if self._p2p is not None and inp.dtype == torch.float32:
            return self._timed("p2p_reduce_scatter_table_grad", lambda: self._p2p.reduce_scatter_avg(out, inp))
        self._timed("reduce_scatter_table_grad", lambda: reduce_scatter_avg(out, inp, self.rank, self.world, self.group))

    def _all_gather(self, store, sl):
        from .dist import all_gather_shards
        if self._comm_stub:
            return store
        if self._p2p is not None:
            name = "table" if store is self.table_store else ("copy16" if store is self.copy16_store else None)
            if name is not None:
                return self._timed("p2p_all_gather_table", lambda: self._p2p.all_gather(name, sl))
        self._timed("all_gather_table", lambda: all_gather_shards(store, self.rank, self.shard_len, self.world, self.group))

    def sync_master(self):
        """Sharded optimizer with a 16-bit table copy: only the copy is exchanged every step; gather the fp32 master table of
        all shards (for a checkpoint / model.state_dict()).  No-op otherwise."""
        self.finish_comm()
        if self.shard and self._master_stale:
            if self._groups is not None:
                self._gather_groups(self.table_store)
            else:
                lo = self.rank * self.shard_len
                self._all_gather(self.table_store, slice(lo, lo + self.shard_len))
            self._master_stale = False

    def _all_reduce(self):
        """Average the gradients of the ray shards (MSE is a mean over the local shard) and OR the inf flags: ONE
        collective over the flat bucket [table grad | MLP grad | flag] (the flag rides along as a float; any rank's
        non-zero flag leaves a non-zero sum / mean)."""
        flag_i = self.state_i[_SI_FOUND_INF:_SI_FOUND_INF + 1]
        self._flag_f.copy_(flag_i)
        if self.half:                                              # the f16 table gradient travels as it is (22.8 MB)
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(self.table_grad, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.table_grad, op=dist.ReduceOp.SUM, group=self.group)
                self.table_grad.div_(self.world)
        buf = self.grad_flat
        if self._comm is not None and not self.shard:
            buf = self._comm
            buf.copy_(self.grad_flat)                           # loss-scaled gradients: bf16 keeps the fp32 exponent range
        def flat():
            if self._comm_stub:
                return
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                buf.div_(self.world)
        self._timed("all_reduce_flat_bucket", flat)
        if buf is not self.grad_flat:
            self.grad_flat.copy_(buf)
        flag_i.copy_(self._flag_f != 0)
        self._flag_f.zero_()

    def step(self, rays_o, rays_d, target, prefetch=None, noise=None):
        """rays_o, rays_d, target: [N,3] contiguous float32 device tensors (this rank's shard).  Returns the per-step
        outputs (device tensors; nothing is synchronised).
        prefetch=(next_rays_o, next_rays_d): the NEXT step's rays, if already known and if the occupancy grid will not be
        updated in between -- their march then overlaps this step on a side stream (the tensors must stay alive and
        unmodified until that next step() call, which must receive the very same tensors)."""
        src, src_next = (rays_o, rays_d), None
        if prefetch is not None:
            src_next = (prefetch[0], prefetch[1])
            prefetch = (prefetch[0].contiguous().float(), prefetch[1].contiguous().float())
        self.stats = self._launch(rays_o.contiguous().float(), rays_d.contiguous().float(), target.contiguous().float(), prefetch,
                                  src, src_next, noise)
        if self._pending_comm and prefetch is None:
            # overlapped exchange: the all-gathers of this step are waited for at the start of the NEXT step -- fine inside a training
            # loop that names its next batch, but a caller that does not (the last step before an evaluation render through
            # model(), a checkpoint through model.state_dict()) must not read a table that is still arriving
            self.finish_comm()
        return self.stats

    def compute_gradients(self, rays_o, rays_d, target, noise=None):
        """Forward + backward of one batch WITHOUT the optimizer (diagnostics / gradient tests): returns the outputs plus
        clones of the table and MLP gradients divided by the current loss scale; accumulators and the inf flag are cleared."""
        if self._graph is not None:
            raise RuntimeError("not available after capture()")
        self._grads_only = True
        try:
            out = self._launch(rays_o.contiguous().float(), rays_d.contiguous().float(), target.contiguous().float(), noise=noise)
        finally:
            self._grads_only = False
        inv = 1.0 / self.state_f[_SF_LOSS_SCALE]
        out["table_grad"], out["mlp_grad"] = self.table_grad.float() * inv, self.mlp_grad * inv
        out["found_inf"] = self.state_i[_SI_FOUND_INF].clone()
        self.grad_flat.zero_()
        if self.half:
            self.table_grad.zero_()
        self.state_i[_SI_FOUND_INF] = 0
        return out

    def capture(self, n_rays):
        """Switch to hipGraph replay (single-GPU; the RCCL path stays eager).  The encode -> MLPs -> composite + loss -> backward
        -> optimizer chain becomes ONE graph launch per step (one graph per march-set parity, captured on first use) preceded by
        a one-launch staging copy of the target colours; the march stays eager -- the current batch's if it was not prefetched,
        the next batch's on the side stream -- because hipGraph executes parallel branches of one graph back to back
        (measured: a march branch inside the graph added its full 0.1 ms to the step), which would undo the overlap."""
        if self.world > 1:
            raise RuntimeError("graph capture is only wired for the single-GPU step")
        TrainArena.get(self.dev, n_rays, self.max_samples)                 # allocate the arena outside the graph pools
        self._march_sets(n_rays)
        if self.chunked:
            # the chunked forward's list counters and per-ray transmittance state: allocated HERE, not lazily inside the first captured
            # step (they would come from graph 0's private pool, graph 1 would use memory it does not own, and the zero-fill would be
            # replayed every other step: ADVICE r5)
            self._chunk_state(n_rays)
        self._static_target = torch.zeros(n_rays, 3, device=self.dev, dtype=torch.float32)
        self._graph = {}                                                    # march-set parity -> (CUDAGraph, outputs)
        self._graph_n = n_rays
        return self._graph

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def update_density_grid(self, density_threshold, warmup=False, **kw):
        """Occupancy-grid maintenance (train.py:178-182).  With world_size > 1 rank 0's freshly updated grid + bitfield are
        broadcast (8.4 MB + 262 KB per cascade, once per 16 steps): the update samples random cells, and replicas that march
        different bitfields stop being replicas."""
        self.finish_comm()
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            self.model.update_density_grid(density_threshold, warmup=warmup, **kw)
        if self.world > 1 and self.sync_occupancy:
            from .dist import broadcast_occupancy
            broadcast_occupancy(self.model, src=0, group=self.group)
        TrainArena.invalidate_coarse()       # the coarse 8^3-block tables are rebuilt on the next march, whoever wrote the bitfield
        for sets in self._sets.values():     # a march prefetched against the old bitfield must not be consumed
            for M in sets:
                M.src = None

    def state_dict(self):
        """Optimizer-side state the model's own state_dict does not hold: Adam moments, loss scale + growth counter, the
        LR-schedule iteration.  (The reference saves no optimizer state either -- ckpt = model.state_dict(), train.py:285-291 --
        so a resume without this restarts the moments and the cosine schedule; with it the continuation is exact.)
        Sharded optimizer: every rank only ever updates the moments of its own 1/world of the table, so the shards are gathered
        here (COLLECTIVE: every rank must call state_dict(), any rank's result is then complete) and the fp32 master table of
        the other shards is brought up to date (sync_master).  The moments are saved WITHOUT the world-dependent padding
        ([:nt]): a checkpoint loads into any world size."""
        self.sync_master()
        if self.shard and self._groups is not None:
            self._gather_groups(self.table_m)
            self._gather_groups(self.table_v)
        elif self.shard:
            lo = self.rank * self.shard_len
            sl = slice(lo, lo + self.shard_len)
            self._all_gather(self.table_m, sl)
            self._all_gather(self.table_v, sl)
        return {"table_m": self.table_m[:self.nt].clone(), "table_v": self.table_v[:self.nt].clone(), "mlp_m": self.mlp_m.clone(),
                "mlp_v": self.mlp_v.clone(), "state_f": self.state_f.clone(), "state_i": self.state_i.clone()}

    def load_state_dict(self, sd):
        """Load what state_dict() returned (on every rank, after loading the model's weights into the model).  Accepts the
        unpadded [:nt] moments of this version and the padded ones of earlier checkpoints written at the same world size."""
        for k in ("table_m", "table_v"):
            src, dst = sd[k], getattr(self, k)
            if src.numel() not in (self.nt, dst.numel()):
                raise ValueError("%s has %d elements; this trainer expects %d (or %d padded)" % (k, src.numel(), self.nt, dst.numel()))
            n = min(src.numel(), self.nt)
            dst[:n].copy_(src[:n])
            dst[self.nt:].zero_()
        for k in ("mlp_m", "mlp_v", "state_f", "state_i"):
            getattr(self, k).copy_(sd[k])
        self.grad_flat.zero_()
        if self.half:
            self.table_grad.zero_()
        if self.shard_grad is not None:
            self.shard_grad.zero_()
        TrainArena.invalidate_coarse()
        # the model's weights were loaded alongside: the fp32 master IS the checkpoint now, on every shard -- nothing left to gather
        self._master_stale = False
        self.repack()                          # refresh the fp16 MFMA image and the 16-bit table copy from the master

    def close(self):
        """Release what the trainer holds outside torch's allocator: the low-priority side stream (ngp_stream_create_low_priority).
        Idempotent; also called when the trainer is garbage-collected."""
        side, self._side, self._side_low = getattr(self, "_side_low", None), None, None
        host, self._marched_host = getattr(self, "_marched_host", None), None
        dflt, self._side_default = getattr(self, "_side_default", None), None
        if host is not None:                  # no copy into it may be in flight when it goes
            try:
                for s_ in (side, dflt):
                    if s_ is not None:
                        s_.synchronize()
                self.L.ngp_host_free(host)
            except Exception:
                pass
        self._adaptive_prefetch = False
        if side is not None and getattr(self, "_side_prio", None) is not None:
            try:
                side.synchronize()
                self.L.ngp_stream_destroy(ctypes.c_void_p(side.cuda_stream))
            except Exception:
                pass
            self._side_prio = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_loss(self):
        """MSE of the last step (host sync: logging only)."""
        se = self.stats.get("sq_err")
        if se is None and self.distortion_loss_w > 0 and self.stats:
            return float(self.state_f[_SF_LOSS].item())                   # (a caller that ran ngp_mse_loss_grad itself)
        return float("nan") if se is None else float(se.sum().item()) / (3.0 * se.numel())

    def loss_scale(self):
        return float(self.state_f[_SF_LOSS_SCALE].item())

    def counters(self):
        s = self.state_i.tolist()
        return {"iter": s[_SI_ITER], "opt_steps": s[_SI_OPT_STEP], "skipped": s[_SI_SKIPPED]}

    def lr_at(self, it):
        return self.eta_min + (self.lr0 - self.eta_min) * 0.5 * (1 + math.cos(math.pi * min(it, self.t_max) / self.t_max))
