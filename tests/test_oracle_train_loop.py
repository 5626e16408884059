"""The CPU training loop the HIP trainer's trajectory is held to (oracle/train_loop.py, a restatement of train.py:137-201 on the oracle
kernels) checked on its own: its optimizer arithmetic against a hand-written Adam + cosine schedule, its occupancy update against the
reference's formulation evaluated with torch, and that it learns.  CPU only; the GPU comparison is tests/test_gpu_trajectory.py."""
import math

import numpy as np
import torch


def _init(seed=0, table_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    def xavier(o, i):
        b = math.sqrt(6.0 / (i + o))
        return ((torch.rand(o, i, generator=g) * 2 - 1) * b).numpy()
    weights = [xavier(64, 32), xavier(16, 64), xavier(64, 32), xavier(64, 64), xavier(3, 64)]
    table = (torch.rand(5710032 * 2, generator=g) * table_scale).numpy()
    return weights, table


def _batch(n, seed):
    import sys, os
    from ngp_hip import synthetic
    o, d = synthetic.lego_rays(n, seed=seed)
    target = synthetic.procedural_render_gt(torch.from_numpy(o), torch.from_numpy(d)).numpy()
    noise = np.random.default_rng(seed).random(n, dtype=np.float32)
    return o, d, target, noise


def test_update_density_grid_matches_reference_formulation(oracle):
    """networks.py:255-290 written with torch exactly as the reference does (coords / (G-1) * 2 - 1 etc. on float tensors) gives the
    grid and bitfield the loop's numpy restatement gives, on the same jitter."""
    from oracle.train_loop import OracleTrainer
    weights, table = _init(1, 1e-2)
    tr = OracleTrainer(weights, table, lr=1e-2, max_steps=64)
    G, G3 = 128, 128**3
    u = np.random.default_rng(3).random((G3, 3), dtype=np.float32)
    mean = tr.update_density_grid(0.01 * 1024 / 3**0.5, [u])
    # reference formulation: grid_coords in ANY enumeration + indices = morton3D(coords); here meshgrid order
    r = torch.arange(G, dtype=torch.int32)
    coords = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    indices = torch.from_numpy(oracle.morton3d(coords.numpy())).long()
    s = min(2**(0 - 1), 0.5)
    hgs = s / G
    xyzs_w = (coords / (G - 1) * 2 - 1) * (s - hgs)
    xyzs_w += (torch.from_numpy(u)[indices] * 2 - 1) * hgs                  # the jitter row belongs to the cell's Morton code
    tmp = torch.zeros(1, G3)
    tmp[0, indices] = torch.from_numpy(tr.density(xyzs_w.numpy(), autocast=True))
    grid = torch.where(torch.zeros(1, G3) < 0, torch.zeros(1, G3), torch.maximum(torch.zeros(1, G3) * 0.95, tmp))
    m_ref = grid[grid > 0].mean().item()
    assert abs(m_ref - mean) <= 1e-5 * mean
    np.testing.assert_array_equal(grid.numpy(), tr.density_grid)
    bits = oracle.packbits(grid.reshape(-1).numpy(), min(mean, 0.01 * 1024 / 3**0.5))
    np.testing.assert_array_equal(bits, tr.bits)
    frac = np.unpackbits(tr.bits).mean()
    assert 0.2 < frac < 0.8            # threshold = mean density: about half the cells (SURVEY section 8 header, networks.py:286-290)


def test_loop_optimizer_is_adam_eps1e15_with_cosine_lr(oracle):
    """Two steps of the loop against Adam + CosineAnnealingLR written out by hand on the gradients the loop left behind."""
    from oracle.train_loop import OracleTrainer
    weights, table = _init(2, 1e-2)
    T, lr0 = 8, 1e-2
    tr = OracleTrainer(weights, table, lr=lr0, max_steps=T)
    tr.bits = np.full_like(tr.bits, 255)
    o, d, target, noise = _batch(96, 5)
    p = [np.array(table, np.float64)] + [np.array(w, np.float64).reshape(-1) for w in weights]
    m = [np.zeros_like(x) for x in p]; v = [np.zeros_like(x) for x in p]
    for k in range(2):
        rec = tr.forward_backward(o, d, target, noise)
        grads = [tr.table.grad.numpy().astype(np.float64)] + [w.grad.numpy().reshape(-1).astype(np.float64) for w in tr.w]
        lr = lr0 / 30 + (lr0 - lr0 / 30) * 0.5 * (1 + math.cos(math.pi * k / T))
        assert abs(tr.opt.param_groups[0]["lr"] - lr) < 1e-12
        tr.opt.step(); tr.sched.step()
        for j, g in enumerate(grads):
            m[j] = 0.9 * m[j] + 0.1 * g
            v[j] = 0.999 * v[j] + 0.001 * g * g
            p[j] = p[j] - lr * (m[j] / (1 - 0.9**(k + 1))) / (np.sqrt(v[j] / (1 - 0.999**(k + 1))) + 1e-15)
        assert np.isfinite(rec["loss"]) and rec["rm_samples"] > 0
    got = [tr.table.detach().numpy()] + [w.detach().numpy().reshape(-1) for w in tr.w]
    for a, b in zip(got, p):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-7)
    assert (got[0] != np.asarray(table, np.float32)).sum() > 1000          # the table really moved (Adam: by ~lr where touched)


def test_loop_learns_and_half_kind_tracks_f32(oracle):
    """12 steps from one initialisation, fp32 and half2 encoders: the loss falls, and the half2 loop (f16 table copy, f16 embedding,
    f16 scaled output gradient) stays close to the fp32 one."""
    from oracle.train_loop import OracleTrainer
    weights, table = _init(3, 1e-2)
    n = 256
    batches = [_batch(n, 60 + k) for k in range(3)]
    u = np.random.default_rng(9).random((128**3, 3), dtype=np.float32)
    curves = {}
    for kind in ("f32", "half"):
        tr = OracleTrainer(weights, table, lr=1e-2, max_steps=12, kind=kind, loss_scale=2.0**10)
        tr.update_density_grid(0.01 * 1024 / 3**0.5, [u])
        curves[kind] = [tr.step(*batches[i % 3])["loss"] for i in range(12)]
    for kind, c in curves.items():
        assert all(np.isfinite(c)) and c[-1] < 0.7 * c[0], (kind, c)
    np.testing.assert_allclose(curves["half"], curves["f32"], rtol=5e-2)
