"""bf16-stored hash table (BASELINE config 2 wording; not in the reference, which has fp32 and fp16 tables).

Semantics pinned here: the forward gathers from a bf16 (round-to-nearest-even) copy of the fp32 master table and
interpolates / outputs in f32, so it must equal -- BIT FOR BIT -- the fp32 kernel and the oracle run on the bf16-rounded
table; gradient, optimizer state and checkpoint stay fp32; the Adam pass keeps the copy current."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16)


@pytest.mark.parametrize("n", [1000, 20000])          # generic kernel / XCD-partitioned kernel
@pytest.mark.parametrize("max_res", [1024, 4096])
def test_fwd_bf16_bit_exact_vs_rounded_f32(oracle, hip_lib, n, max_res):
    from ngp_hip import ops
    lv = ops.make_levels(2**19, 16, 16, max_res, 2)
    rng = np.random.default_rng(3)
    x = rng.random((n, 3), dtype=np.float32)
    x[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.999999, 1e-7, 0.5]]
    table = torch.from_numpy((rng.standard_normal(lv.total_entries * 2) * 3).astype(np.float32)).cuda()
    t16 = ops.cast_bf16(table)
    assert torch.equal(_bits(t16), _bits(table.bfloat16()))                  # RNE, same as torch
    rounded = t16.float()
    got = ops.hash_fwd_bf16(torch.from_numpy(x).cuda(), t16, lv)
    via_f32 = ops.hash_fwd_f32(torch.from_numpy(x).cuda(), rounded, lv)
    assert torch.equal(_bits(got), _bits(via_f32))
    ref = oracle.hash_fwd_f32(x, rounded.cpu().numpy(), oracle.make_levels(2**19, 16, 16, max_res, 2))
    assert np.array_equal(got.cpu().numpy().view(np.int32), ref.view(np.int32))


def test_cast_bf16_special_values(hip_lib):
    from ngp_hip import ops
    v = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.001953125, 1.005859375, 3.4e38, -3.4e38, float("inf"), float("-inf"),
                      1e-40, float("nan")], device="cuda")
    got, ref = ops.cast_bf16(v), v.bfloat16()
    assert torch.equal(_bits(got)[:-1], _bits(ref)[:-1]) and torch.isnan(got[-1].float())


def test_adam_step_bf16_keeps_copy_current(hip_lib):
    import ctypes
    from ngp_hip import lib as L
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    n = 1 << 16
    torch.manual_seed(0)
    p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1024
    g[:4096] = 0                                   # untouched entries: skipped float4s keep the (initial) copy
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p2, g2, m2, v2 = p.clone(), g.clone(), m.clone(), v.clone()
    shadow = p.bfloat16()
    sf = torch.zeros(8, device="cuda"); si = torch.zeros(8, device="cuda", dtype=torch.int32)
    sf[0] = 1024.0
    L.check(lib.ngp_train_prologue(_ptr(sf), _ptr(si), 1e-2, 1e-2 / 30, 100, 0.9, 0.999, 2.0, 0.5, 2000, _stream()), "prologue")
    L.check(lib.ngp_adam_step_bf16(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, _ptr(shadow),
                                   _stream()), "adam_bf16")
    L.check(lib.ngp_adam_step(_ptr(p2), _ptr(g2), _ptr(m2), _ptr(v2), n, _ptr(sf), _ptr(si), 0.9, 0.999, 1e-15, _stream()), "adam")
    torch.cuda.synchronize()
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and not g.any()
    assert torch.equal(_bits(shadow), _bits(p.bfloat16()))
    assert (p[4096:] != shadow[4096:].float()).any()           # the copy really is the rounded value, not the master


def test_module_bf16_forward_backward_and_refresh(hip_lib):
    from modules.hash_encoder import HashEncoder
    torch.manual_seed(1)
    enc = HashEncoder(max_res=1024, table_dtype=torch.bfloat16).cuda()
    ref = HashEncoder(max_res=1024).cuda()
    assert set(enc.state_dict()) == set(ref.state_dict()) and enc.hash_table.dtype == torch.float32
    with torch.no_grad():
        ref.hash_table.copy_(enc.hash_table.bfloat16().float())
    x = torch.rand(5000, 3, device="cuda")
    ya, yb = enc(x), ref(x)
    assert torch.equal(ya, yb)
    w = torch.randn_like(ya)
    (ya * w).sum().backward(); (yb * w).sum().backward()
    # same scatter-add kernel, float atomics: order-nondeterministic
    torch.testing.assert_close(enc.hash_table.grad, ref.hash_table.grad, rtol=1e-5, atol=1e-5)
    # a torch optimizer step moves the parameter's version counter -> the copy is re-cast on the next forward
    torch.optim.SGD(enc.parameters(), lr=0.5).step()
    yc = enc(x)
    assert torch.equal(_bits(enc.table_bf16()), _bits(enc.hash_table.detach().bfloat16()))
    assert not torch.equal(ya, yc)


def test_trainer_bf16_table(hip_lib, lego_bitfield):
    from modules.networks import NGP
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024, table_dtype=torch.bfloat16).cuda()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.2)
    m32 = NGP(scale=0.5, max_res=1024).cuda()
    m32.load_state_dict(m.state_dict())                      # same checkpoint keys / shapes
    n = 4096
    o, d = synthetic.lego_rays(n, seed=9)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = torch.rand(n, 3, device="cuda") * 0.5 + 0.25
    tr, tr32 = FusedTrainer(m, max_steps=200), FusedTrainer(m32, max_steps=200)
    assert tr.table_bf16 is not None and tr32.table_bf16 is None
    losses, losses32 = [], []
    for i in range(40):
        torch.manual_seed(50 + i); tr.step(o, d, target); losses.append(tr.last_loss())
        torch.manual_seed(50 + i); tr32.step(o, d, target); losses32.append(tr32.last_loss())
    torch.cuda.synchronize()
    assert torch.equal(_bits(tr.table_bf16), _bits(m.pos_encoder.hash_table.detach().bfloat16()))     # copy stays current
    assert losses[-1] < 0.8 * losses[0]                     # random targets: the floor is their variance
    # bf16 storage of the parameters perturbs the forward by <= 2^-9 relative: the loss curves stay together
    assert abs(losses[-1] - losses32[-1]) < 0.05 * losses32[-1] + 1e-4
    # the occupancy update reads the same bf16 copy
    tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=True)
    torch.cuda.synchronize()
    assert torch.isfinite(m.density_grid).all()


def test_gradient_transport_defaults_to_the_table_copys_width(hip_lib):
    """Round 6: a trainer whose model already reads a bf16 storage copy of the table exchanges its gradient as
    bf16 by default (the parameters come back as that copy too); an fp32 table keeps the exact fp32 mean; explicit values pin it."""
    from modules.networks import NGP
    from ngp_hip.trainer import FusedTrainer
    for table_dtype, want in ((torch.bfloat16, torch.bfloat16), (None, torch.float32)):
        tr = FusedTrainer(NGP(scale=0.5, max_res=1024, table_dtype=table_dtype).cuda())
        assert tr.grad_comm_dtype == want
        tr.close()
    tr = FusedTrainer(NGP(scale=0.5, max_res=1024, table_dtype=torch.bfloat16).cuda(), grad_comm_dtype=torch.float32)
    assert tr.grad_comm_dtype == torch.float32
    tr.close()
