"""Round 5: the table's optimizer in the scatter-add's flush (ngp_hash_bwd_sliced_main_adam, csrc/hash_bwd_lds.hip) against the
two-launch path it replaces (ngp_hash_bwd_sliced_main[_slabs] -> ngp_train_prologue -> ngp_adam_all_ex over the whole table;
reference: train.py:197-201 on the gradient modules/hash_encoder.py:269 produces).

Kernel level, through the C ABI: both paths start every step from the SAME state and the same inputs.  The levels the flush owns
(non-replicated slices: the hashed levels) must come out bit-identical -- parameters, both moments, the bf16 copy; the replicated
coarse levels go through the gradient table and the ordinary optimizer launch in both paths (float atomics between replicas: compared
to 1e-6), and the MLP block of that launch is deterministic (bit-identical).  The sequence holds a forced overflow step (the
GradScaler skip), a step without any live sample (the moments still decay) and steps whose gradients leave most entries untouched.
"""
import ctypes

import numpy as np
import pytest
import torch

from ngp_hip import ops

pytestmark = pytest.mark.gpu

LR0, ETA_MIN, T_MAX, B1, B2, EPS = 1e-2, 1e-2 / 30, 20000, 0.9, 0.999, 1e-15
GROWTH, BACKOFF, GROWTH_INTERVAL = 2.0, 0.5, 7            # a short interval: the scale grows inside the sequence
N_MLP = 9408


def _points(rng, n):
    """Ray-like runs (consecutive samples share coarse cells) + uniform points."""
    if n == 0:
        return np.zeros((0, 3), np.float32)
    n_rays = max(1, n // 48)
    o = rng.random((n_rays, 1, 3), dtype=np.float32) * 0.8 + 0.1
    d = rng.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(40, dtype=np.float32) * np.float32(0.002))[None, :, None]
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3)
    x = np.concatenate([x, rng.random((max(0, n - x.shape[0]), 3), dtype=np.float32)])[:n]
    return np.ascontiguousarray(x.astype(np.float32))


class _State:
    """Everything one optimisation step reads and writes."""

    def __init__(self, lv, copy, gen):
        nt = lv.total_entries * 2
        f32 = dict(device="cuda", dtype=torch.float32)
        self.table = (torch.rand(nt, generator=gen, **f32) * 2 - 1) * 1e-4
        self.m, self.v, self.g = torch.zeros(nt, **f32), torch.zeros(nt, **f32), torch.zeros(nt, **f32)
        self.copy = self.table.to(torch.bfloat16) if copy else None
        self.mlp = torch.randn(N_MLP, generator=gen, **f32) * 0.1
        self.mlp_m, self.mlp_v, self.mlp_g = torch.zeros(N_MLP, **f32), torch.zeros(N_MLP, **f32), torch.zeros(N_MLP, **f32)
        self.sf = torch.zeros(8, **f32)
        self.si = torch.zeros(8, device="cuda", dtype=torch.int32)
        self.sf[0] = 1024.0
        self.wpack = torch.zeros(ops._lib().ngp_mlp_wpack_halfs(), device="cuda", dtype=torch.float16)

    def clone(self):
        o = object.__new__(_State)
        for k, t in self.__dict__.items():
            setattr(o, k, None if t is None else t.clone())
        return o


def _prologue(L, S):
    assert L.ngp_train_prologue(ops._ptr(S.sf), ops._ptr(S.si), LR0, ETA_MIN, T_MAX, B1, B2, GROWTH, BACKOFF, GROWTH_INTERVAL,
                                ops._stream()) == 0


def _bits(t):
    return t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16)


@pytest.mark.parametrize("fold", [False, True], ids=["prologue-launch", "prologue-in-scatter"])
@pytest.mark.parametrize("copy", [False, True], ids=["f32", "bf16copy"])
def test_flush_adam_bit_identical_to_two_launch_path(hip_lib, copy, fold):
    """fold: ngp_hash_bwd_sliced_main_adam_step -- the GradScaler / schedule decision evaluated inside the scatter-add launch (every
    workgroup on a private copy of the state, the last one out publishes it) instead of by ngp_train_prologue in front of it."""
    L = ops._lib()
    lv = ops.make_levels(2**19, 16, 16, 1024, 2)                    # the C2 table
    nt = lv.total_entries * 2
    prefix = int(L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lv)))
    assert 0 < prefix < nt and prefix % 4 == 0
    # the prefix is exactly the levels whose slices are replicated over sample ranges
    nrep = (ctypes.c_uint8 * 16)()
    assert L.ngp_hash_bwd_sliced_plan(ctypes.byref(lv), None, 0, None, None, nrep, None, None) > 0
    first = min(l for l in range(16) if nrep[l] == 1)
    assert all(nrep[l] == 1 for l in range(first, 16)) and prefix == lv.offset[first] * 2

    gen = torch.Generator(device="cuda").manual_seed(5)
    rng = np.random.default_rng(5)
    S = _State(lv, copy, gen)
    cap = 60000
    ws = ops.sliced_workspace(lv, cap, torch.device("cuda"))
    x_buf = torch.zeros(cap, 3, device="cuda")
    d_buf = torch.zeros(cap, 32, device="cuda")
    kind = 1 if copy else 0
    n_steps, overflow_at, empty_at = 52, (9, 30), 17
    skipped = 0
    for step in range(n_steps):
        n = 0 if step == empty_at else int(rng.integers(2000, cap))
        if step % 5 == 4:
            n = int(rng.integers(1, 300))                          # a handful of samples: almost every entry has g = 0
        x = _points(rng, n)
        x_buf[:n] = torch.from_numpy(x).cuda()
        scale = float(S.sf[0])
        d_buf[:n] = torch.from_numpy((rng.standard_normal((n, 32)) * np.exp(rng.uniform(-12, 2, (n, 1)))).astype(np.float32)).cuda() * scale
        if n:
            d_buf[:n:7] = 0.0
        cnt = torch.tensor([n], device="cuda", dtype=torch.int32)
        parts = torch.randn(3 * N_MLP, device="cuda", generator=gen) * scale * 1e-3      # three weight-gradient slabs
        if step in overflow_at:
            S.si[3] = 1                                            # what the MLP backward raises on a non-finite d_enc / dW

        def prep():
            assert L.ngp_hash_bwd_sliced_prep(ops._ptr(x_buf), ctypes.byref(lv), cap, ops._ptr(cnt), ops._ptr(None), 0, 0.0, 1.0,
                                              ops._ptr(ws), ws.numel(), ops._stream()) == 0

        # ---- A: scatter-add into the gradient table, prologue, optimizer over the whole table
        A = S.clone()
        prep()
        assert L.ngp_hash_bwd_sliced_main_slabs(ops._ptr(d_buf), ctypes.byref(lv), cap, ops._ptr(cnt), 0, ops._ptr(A.g), 0, ops._ptr(None),
                                                ops._ptr(ws), ws.numel(), ops._ptr(parts), 3, ops._ptr(A.mlp_g), ops._stream()) == 0
        _prologue(L, A)
        assert L.ngp_adam_all_ex(ops._ptr(A.table), ops._ptr(A.g), 0, ops._ptr(A.m), ops._ptr(A.v), nt, ops._ptr(A.copy), kind,
                                 ops._ptr(A.mlp), ops._ptr(A.mlp_g), ops._ptr(A.mlp_m), ops._ptr(A.mlp_v), ops._ptr(A.sf), ops._ptr(A.si),
                                 B1, B2, EPS, 0, ops._ptr(A.wpack), ops._stream()) == 0
        # ---- B: prologue, scatter-add with the optimizer in its flush, optimizer over the replicated levels only
        Bs = S.clone()
        prep()
        if fold:
            assert L.ngp_hash_bwd_sliced_main_adam_step(ops._ptr(d_buf), ctypes.byref(lv), cap, ops._ptr(cnt), 0, ops._ptr(Bs.g), ops._ptr(ws),
                                                        ws.numel(), ops._ptr(parts), 3, ops._ptr(Bs.mlp_g), ops._ptr(Bs.table),
                                                        ops._ptr(Bs.m), ops._ptr(Bs.v), ops._ptr(Bs.copy), ops._ptr(Bs.sf), ops._ptr(Bs.si),
                                                        LR0, ETA_MIN, T_MAX, B1, B2, EPS, GROWTH, BACKOFF, GROWTH_INTERVAL,
                                                        ops._stream()) == 0
        else:
            _prologue(L, Bs)
            assert L.ngp_hash_bwd_sliced_main_adam(ops._ptr(d_buf), ctypes.byref(lv), cap, ops._ptr(cnt), 0, ops._ptr(Bs.g), ops._ptr(ws),
                                                   ws.numel(), ops._ptr(parts), 3, ops._ptr(Bs.mlp_g), ops._ptr(Bs.table), ops._ptr(Bs.m),
                                                   ops._ptr(Bs.v), ops._ptr(Bs.copy), ops._ptr(Bs.sf), ops._ptr(Bs.si), B1, B2, EPS,
                                                   ops._stream()) == 0
        # the flush-owned levels never touch the gradient table
        assert float(Bs.g[prefix:].abs().max()) == 0.0
        assert L.ngp_adam_all_ex(ops._ptr(Bs.table), ops._ptr(Bs.g), 0, ops._ptr(Bs.m), ops._ptr(Bs.v), prefix, ops._ptr(Bs.copy), kind,
                                 ops._ptr(Bs.mlp), ops._ptr(Bs.mlp_g), ops._ptr(Bs.mlp_m), ops._ptr(Bs.mlp_v), ops._ptr(Bs.sf),
                                 ops._ptr(Bs.si), B1, B2, EPS, 0, ops._ptr(Bs.wpack), ops._stream()) == 0
        torch.cuda.synchronize()
        tag = "step %d (n = %d)" % (step, n)
        assert torch.equal(A.si, Bs.si) and torch.equal(_bits(A.sf), _bits(Bs.sf)), tag
        is_skip = int(A.si[4]) != 0
        assert is_skip == (step in overflow_at), tag
        skipped += int(is_skip)
        for name in ("table", "m", "v") + (("copy",) if copy else ()):
            a, b = getattr(A, name), getattr(Bs, name)
            assert torch.equal(_bits(a[prefix:]), _bits(b[prefix:])), "%s: %s differs on the flush-owned levels" % (tag, name)
            # (replicas of a coarse slice meet in the gradient table with float atomics: ~1e-7 of the gradient, in both paths)
            # (the 16-bit copy of such an entry can land on the other side of a rounding boundary: one bf16 ulp = 2^-8 relative)
            ref_ = a[:prefix].float().cpu().numpy()
            np.testing.assert_allclose(b[:prefix].float().cpu().numpy(), ref_, rtol=2.0**-7 if name == "copy" else 1e-4,
                                       atol=1e-6 * float(np.abs(ref_).max()) + 1e-30,
                                       err_msg="%s: %s, replicated levels" % (tag, name))
        for name in ("mlp", "mlp_m", "mlp_v", "wpack"):
            assert torch.equal(_bits(getattr(A, name)), _bits(getattr(Bs, name))), "%s: %s" % (tag, name)
        assert float(A.g.abs().max()) == 0.0 and float(Bs.g.abs().max()) == 0.0 and float(Bs.mlp_g.abs().max()) == 0.0, tag
        if is_skip:                                                # a skipped step leaves parameters and moments alone
            assert torch.equal(_bits(S.table), _bits(Bs.table)) and torch.equal(_bits(S.m), _bits(Bs.m)), tag
        elif n == 0 and step > 0:                                  # no live sample: the moments decay, the parameters still move
            assert not torch.equal(_bits(S.m[prefix:]), _bits(Bs.m[prefix:])), tag
        S = A
    assert skipped == len(overflow_at)
    assert int(S.si[1]) == n_steps - skipped                       # optimizer steps taken
    assert float(S.sf[0]) != 1024.0                                # the loss scale moved (growth and back-off both happened)
    if copy:
        assert torch.equal(S.copy, S.table.to(torch.bfloat16))


def test_flush_adam_refused_where_no_level_qualifies(hip_lib):
    """A table whose hashed levels have fewer than 64 slices is replicated on every level: the entry points say -2 and the caller
    keeps the two-launch path."""
    L = ops._lib()
    lv = ops.make_levels(2**15, 16, 16, 512, 2)
    assert int(L.ngp_hash_bwd_sliced_adam_prefix(ctypes.byref(lv))) == -2


# ---- trainer level -----------------------------------------------------------------------------------------------------------------
def _trainer(lego_bitfield, n, flush, det, table_dtype=None, seed=0):
    from modules.networks import NGP
    from ngp_hip.trainer import FusedTrainer
    torch.manual_seed(seed)
    m = NGP(scale=0.5, max_res=1024, table_dtype=table_dtype).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    tr = FusedTrainer(m, lr=1e-2, max_steps=2000, init_scale=2.0**12, growth_interval=11)
    tr._flush_adam = flush
    tr.set_deterministic(det)
    return m, tr


def _batches(n, k):
    from ngp_hip import synthetic
    out = []
    for b in range(k):
        o, d = synthetic.lego_rays(n, seed=300 + b)
        o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        out.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
    return out


def _run(tr, m, pool, steps, poison_at=(), update_at=()):
    thr = 0.01 * 1024 / 3**0.5
    for i in range(steps):
        if i in update_at:
            tr.update_density_grid(thr, warmup=(i == update_at[0]))
        o, d, tgt = pool[i % len(pool)]
        if i in poison_at:
            tgt = tgt.clone(); tgt[5, 1] = float("nan")             # -> non-finite d_enc / dW: the GradScaler skip
        nxt = pool[(i + 1) % len(pool)]
        tr.step(o, d, tgt, prefetch=None if (i + 1) in update_at else (nxt[0], nxt[1]))
    torch.cuda.synchronize()


def _state_bits(m, tr):
    out = {"table": tr.table, "m": tr.table_m, "v": tr.table_v, "mlp": tr.mlp_flat, "mlp_m": tr.mlp_m, "mlp_v": tr.mlp_v,
           "grid": m.density_grid, "bits": m.density_bitfield, "sf": tr.state_f, "si": tr.state_i}
    if tr.copy16_store is not None:
        out["copy16"] = tr.copy16_store
    return {k: v.detach().clone() for k, v in out.items()}


def _same(a, b):
    return torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b)


@pytest.mark.parametrize("table_dtype", [None, torch.bfloat16], ids=["f32", "bf16copy"])
def test_trainer_flush_adam_equals_two_launch_path(hip_lib, lego_bitfield, table_dtype):
    """FusedTrainer with the optimizer in the scatter-add's flush (the default) against the same trainer with NGP_EXPERIMENT flush_adam=0
    semantics, 56 steps on a scene: one forced overflow step, an all-cell and a sampled occupancy update.  Both run in deterministic
    mode, so the comparison is bit for bit on everything the step writes (table, both moments, 16-bit copy, MLP, occupancy grid,
    GradScaler / schedule state)."""
    n = 2048
    pool = _batches(n, 4)
    res = []
    for flush in (True, False):
        m, tr = _trainer(lego_bitfield, n, flush, True, table_dtype)
        torch.manual_seed(77)
        _run(tr, m, pool, 56, poison_at=(21,), update_at=(16, 32, 48))
        assert tr.counters()["skipped"] >= 1 and tr.counters()["opt_steps"] >= 50
        assert (tr._adam_prefix.get(1) == 0) == flush                 # deterministic plan: every level has one owner per slice
        res.append(_state_bits(m, tr))
    for k in res[0]:
        assert _same(res[0][k], res[1][k]), k
    assert float(res[0]["table"].abs().max()) > 0


def test_trainer_deterministic_mode_reproduces_itself(hip_lib, lego_bitfield):
    """What bench.py's pinned conditioning relies on: two trainers built from the same seed and stepped in deterministic mode hold
    bit-identical state after 40 steps (incl. occupancy updates) -- and the default fast path, given that state, produces the same
    sample counts within a fraction of a per cent."""
    n = 2048
    pool = _batches(n, 4)
    res, live = [], []
    for _ in range(2):
        m, tr = _trainer(lego_bitfield, n, True, True)
        torch.manual_seed(78)
        _run(tr, m, pool, 40, update_at=(0, 16, 32))
        res.append(_state_bits(m, tr))
        tr.set_deterministic(False)                                    # ... then the default path on top of the pinned state
        _run(tr, m, pool, 6)
        live.append(int(tr._live_total[0]))
    for k in res[0]:
        assert _same(res[0][k], res[1][k]), k
    assert abs(live[0] - live[1]) <= 0.005 * max(live) + 2, live
