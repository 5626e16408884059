"""End-to-end parity on a model WITH CONTENT: the assembled HIP pipeline against the oracle pipeline on a
conditioned model -- opaque surfaces, early termination on most non-empty rays -- for the fp32, bf16-copy and half2 encoders,
for the C3 (Garden-shape) configuration, and for the end-to-end GRADIENTS (table + five MLP weights) against the fp32 CPU chain
(oracle hash / composite kernels + fp32 torch autograd for the MLPs), with the torch-autocast-vs-fp32 gap as the yardstick.

BASELINE north_star: indexing / compaction bit-exact, rendered radiance within 1e-3."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _h(x):                       # round to fp16, keep computing in fp32 (what an fp16 tensor holds)
    return x.astype(np.float16).astype(np.float32)


def _linear_autocast(x, w):      # torch autocast Linear: fp16 operands, fp32 accumulate, fp16 result
    return _h(_h(x) @ _h(w).T)


def _bf16_round(a):
    return torch.from_numpy(a).bfloat16().float().numpy()


def _oracle_forward(oracle, weights, table, o, d, bits, noise, scale, cascades, esf, bg, max_res, kind="f32", T_thr=1e-4):
    """ray_aabb -> march -> hash encode -> MLPs (numpy, torch-autocast's fp16 rounding points emulated) -> composite."""
    hits = oracle.ray_aabb(o, d, scale)
    rays_a, xyzs, dirs, deltas, ts, total = oracle.march_train(o, d, hits, bits, noise, cascades, scale, esf, 128, 1024)
    lv = oracle.make_levels(2**19, 16, 16, max_res, 2)
    x01 = ((xyzs - (-scale)) / (scale - (-scale))).astype(np.float32)                 # networks.py:144
    if kind == "half":                                                                # hash_encoder_half.py:112-161
        enc = oracle.hash_fwd_f16(x01, table.reshape(-1, 2).astype(np.float16), lv).reshape(-1, 32).astype(np.float32)
    else:
        enc = oracle.hash_fwd_f32(x01, _bf16_round(table) if kind == "bf16" else table, lv)
    W1, W2, W3, W4, W5 = weights
    h = _linear_autocast(np.maximum(_linear_autocast(enc, W1), 0), W2)                # xyz_encoder 32 -> 64 -> 16
    sigmas = np.exp(h[:, 0].astype(np.float32))                                       # TruncExp on h[:, 0], fp32
    assert np.isfinite(sigmas).all()
    dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    sh = oracle.sh16_fwd(((dn + 1) / 2).astype(np.float32))                           # networks.py:162-163
    x = np.concatenate([sh, h], 1)
    x = np.maximum(_linear_autocast(x, W3), 0)
    x = np.maximum(_linear_autocast(x, W4), 0)
    rgbs = _h(1.0 / (1.0 + np.exp(-_linear_autocast(x, W5))))                         # Sigmoid on an fp16 tensor
    vr, op, dep, rgb, ws = oracle.composite_train_fwd(sigmas, rgbs, deltas, ts, rays_a, T_thr)
    rgb = rgb + bg * (1.0 - op)[:, None]                                              # rendering.py:219-226
    return {"rays_a": rays_a, "total": total, "rgb": rgb, "opacity": op, "depth": dep, "vr": vr}


@pytest.fixture(scope="module")
def conditioned(hip_lib):
    """An fp32 model trained for 320 steps on the analytic Lego-shape scene (ngp_hip/synthetic.py), then sharpened: the
    occupancy keeps only cells denser than 10 and the density logit row is scaled by 1.5, so that rays through occupied
    space meet opaque matter and terminate early."""
    from modules.networks import NGP
    from modules.utils import packbits
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    torch.manual_seed(5)
    m = NGP(scale=0.5, max_res=1024).cuda()
    tr = FusedTrainer(m, lr=1e-2, max_steps=2000)
    tr.set_deterministic(True)            # (round 5: the same state on every box and run -- the assertions below sit on its statistics)
    pool = []
    for b in range(8):
        o, d = synthetic.lego_rays(4096, seed=300 + b)
        o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        pool.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
    for i in range(320):
        if i % 16 == 0:
            tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=i < 256)
        tr.step(*pool[i % 8])
    tr.set_deterministic(False)
    with torch.no_grad():
        packbits(m.density_grid.reshape(-1).contiguous(), 10.0, m.density_bitfield)
        # sharpen: scale the density logit row, but keep exp() far from overflow (unsupervised interiors extrapolate)
        from ngp_hip.fused import TrainArena
        out = tr.compute_gradients(*pool[0])
        sig = TrainArena.get(m.density_grid.device, 4096, 1024).sigmas[:int(out["rm_samples"][0])]
        h0_max = float(sig.max().log())
        alpha = min(1.5, 30.0 / max(h0_max, 1e-3))
        m.xyz_encoder.output_layer.weight.data[0] *= alpha
        print("conditioned fixture: max density logit %.1f -> logit row scaled by %.2f" % (h0_max, alpha))
    torch.cuda.synchronize()
    return {"state": copy.deepcopy(m.state_dict()), "bits": m.density_bitfield.cpu().numpy().copy()}


def _model(kind, state):
    from modules.networks import NGP
    m = NGP(scale=0.5, max_res=1024, half_opt=kind == "half", table_dtype=torch.bfloat16 if kind == "bf16" else None).cuda()
    sd = copy.deepcopy(state)
    if kind == "half":
        sd["pos_encoder.hash_table"] = sd["pos_encoder.hash_table"].view(-1, 2)      # hash_encoder_half.py: 2-D parameter
        sd = {k: v for k, v in sd.items() if "hash_grad" not in k}
    m.load_state_dict(sd, strict=False)
    return m


def _noise_for(n, seed):
    torch.manual_seed(seed)
    noise = torch.rand(n, device="cuda").cpu().numpy()          # render(): the first draw after the seed is the march jitter; the
    # trainer draws its jitter in the march kernel (round 4) and is handed this vector explicitly
    torch.manual_seed(seed)
    return noise


@pytest.mark.parametrize("kind", ["f32", "bf16", "half"])
def test_trainer_forward_matches_oracle_on_content(oracle, conditioned, kind):
    """FusedTrainer's forward (the path bench.py measures), all three encoders, against the oracle pipeline."""
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    n = 2048
    m = _model(kind, conditioned["state"])
    tr = FusedTrainer(m, init_scale=2.0**10)
    o, d = synthetic.lego_rays(n, seed=41)
    target = torch.rand(n, 3, device="cuda")
    noise = _noise_for(n, 77)
    out = tr.compute_gradients(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), target, noise=torch.from_numpy(noise).cuda())
    table = m.pos_encoder.hash_table.detach().float().view(-1).cpu().numpy()
    ref = _oracle_forward(oracle, [w.detach().cpu().numpy() for w in m._mlp_weights()], table, o, d, conditioned["bits"], noise,
                          0.5, 1, 0.0, 1.0, 1024, kind)
    rm, vr = int(out["rm_samples"][0]), out["vr_per_ray"].cpu().numpy()
    counts = ref["rays_a"][np.argsort(ref["rays_a"][:, 0]), 2]
    nonempty = counts > 0
    early = (ref["vr"] < counts) & nonempty
    op_hit = ref["opacity"][nonempty].mean()
    print("e2e content [%s]: %d samples, %d non-empty rays, %.0f%% of them terminate early, mean opacity of non-empty rays %.2f, "
          "composited %d of %d samples" % (kind, rm, nonempty.sum(), 100.0 * early.sum() / nonempty.sum(), op_hit, vr.sum(), rm))
    # the state is the one the review asked for: content, and early termination on about half of the non-empty rays (48-55 % over
        # the rounds' runs: the conditioning is 320 steps of a chaotic optimisation)
    assert nonempty.sum() > 200 and op_hit >= 0.5 and early.sum() > 0.4 * nonempty.sum() and vr.sum() < rm
    # indexing / compaction: bit-exact
    assert rm == ref["total"]
    # (ray id and sample count per ray; the ranges' starts depend on the order the fused march kernel's blocks finish in)
    assert np.array_equal(out["rays_a"].cpu().numpy()[:, [0, 2]], ref["rays_a"][:, [0, 2]])
    dv = np.abs(vr.astype(np.int64) - ref["vr"].astype(np.int64))
    # the early-termination sample is decided by T <= 1e-4 on a product of ~10 f32 factors: allow the boundary sample to differ
    # on a handful of rays (summation order of the wave scan), never more than one sample
    assert dv.max() <= 1 and (dv > 0).mean() < 0.01, (dv.max(), (dv > 0).mean())
    # the trainer returns the composited colour; the background blend (white: rendering.py:219-226) happens inside its loss
    got = out["rgb"].cpu().numpy() + 1.0 * (1.0 - out["opacity"].cpu().numpy())[:, None]
    err = np.abs(got - ref["rgb"])
    print("   max |d rgb| %.2e, mean %.2e, rays off by more than 1e-3: %d" % (err.max(), err.mean(), int((err.max(1) > 1e-3).sum())))
    assert err.max() <= 1e-3                              # north_star: rendered radiance within 1e-3
    assert err.mean() < 2e-5
    np.testing.assert_allclose(out["opacity"].cpu().numpy(), ref["opacity"], atol=1e-3)
    np.testing.assert_allclose(out["depth"].cpu().numpy(), ref["depth"], atol=1e-3)


def test_render_operator_and_fused_match_oracle_on_content(oracle, conditioned):
    """modules.rendering.render (what the reference's train.py calls): fused node and operator path."""
    from modules.rendering import render
    from ngp_hip import synthetic
    n = 2048
    m = _model("f32", conditioned["state"])
    o, d = synthetic.lego_rays(n, seed=42)
    table = m.pos_encoder.hash_table.detach().float().view(-1).cpu().numpy()
    for fused in (True, False):
        noise = _noise_for(n, 78)
        os.environ["NGP_FUSED_RENDER"] = "1" if fused else "0"
        try:
            with torch.autocast("cuda", dtype=torch.float16):
                res = render(m, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), exp_step_factor=0.0)
        finally:
            os.environ["NGP_FUSED_RENDER"] = "1"
        ref = _oracle_forward(oracle, [w.detach().cpu().numpy() for w in m._mlp_weights()], table, o, d, conditioned["bits"], noise,
                              0.5, 1, 0.0, 1.0, 1024)
        assert int(res["rm_samples"]) == ref["total"] and np.array_equal(res["rays_a"].cpu().numpy()[:, [0, 2]], ref["rays_a"][:, [0, 2]])
        assert int(res["vr_samples"]) < int(res["rm_samples"])
        err = np.abs(res["rgb"].float().detach().cpu().numpy() - ref["rgb"])
        print("render fused=%s on content: max |d rgb| %.2e" % (fused, err.max()))
        assert err.max() <= 1e-3


def test_c3_garden_shape_matches_oracle(oracle, hip_lib):
    """BASELINE C3: scale 16, 6 cascades, exp_step_factor 1/256, max_res 4096, black background (rendering.py:219-226,
    train.py:54,105) -- the assembled HIP pipeline (FusedTrainer forward) against the oracle pipeline."""
    from modules.networks import NGP
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    n = 4096
    torch.manual_seed(3)
    m = NGP(scale=16.0, max_res=4096).cuda()
    assert m.cascades == 6
    bits = synthetic.ball_slab_bitfield(6, 16.0, seed=23)
    m.density_bitfield.copy_(torch.from_numpy(bits).cuda())
    tr = FusedTrainer(m, exp_step_factor=1.0 / 256, init_scale=2.0**10)
    assert tr.bg == 0.0
    o, d = synthetic.garden_rays(n, seed=9)
    # content: scale the density logit row of the random-init model so that the densest samples reach exp(20) -- rays through
    # the ball / the ground slab then turn opaque within a few samples and terminate early
    from ngp_hip.fused import TrainArena
    probe = tr.compute_gradients(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.rand(n, 3, device="cuda"))
    sig = TrainArena.get(torch.device("cuda", torch.cuda.current_device()), n, 1024).sigmas[:int(probe["rm_samples"][0])]
    h0 = sig.log()
    hi, lo = float(h0.max()), float(h0.min())
    with torch.no_grad():                                                   # (flip the row if the random init only has negative logits)
        m.xyz_encoder.output_layer.weight.data[0] *= (20.0 / hi) if hi >= -lo else (-20.0 / -lo)
    tr.repack()
    target = torch.rand(n, 3, device="cuda")
    noise = _noise_for(n, 79)                                               # (after every other draw: the march jitter comes next)
    out = tr.compute_gradients(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), target, noise=torch.from_numpy(noise).cuda())
    table = m.pos_encoder.hash_table.detach().float().view(-1).cpu().numpy()
    ref = _oracle_forward(oracle, [w.detach().cpu().numpy() for w in m._mlp_weights()], table, o, d, bits, noise, 16.0, 6, 1.0 / 256, 0.0, 4096)
    rm = int(out["rm_samples"][0])
    print("e2e C3: %d samples (%.1f per ray), %d composited, mean opacity %.2f" % (rm, rm / n, int(out["vr_per_ray"].sum()), ref["opacity"].mean()))
    assert rm == ref["total"] > 50 * n // 4
    # (ray id and sample count per ray; the ranges' starts depend on the order the fused march kernel's blocks finish in)
    assert np.array_equal(out["rays_a"].cpu().numpy()[:, [0, 2]], ref["rays_a"][:, [0, 2]])
    assert int(out["vr_per_ray"].sum()) < 0.8 * rm and ref["opacity"].mean() > 0.3       # content; early termination is exercised
    dv = np.abs(out["vr_per_ray"].cpu().numpy().astype(np.int64) - ref["vr"].astype(np.int64))
    assert dv.max() <= 1 and (dv > 0).mean() < 0.01
    err = np.abs(out["rgb"].cpu().numpy() - ref["rgb"])              # black background: nothing to blend
    print("   max |d rgb| %.2e" % err.max())
    assert err.max() <= 1e-3


def _cpu_fp32_gradients(oracle, weights, table, o, d, bits, noise, target):
    """The CPU chain of bench.py's cpu_baseline, as a gradient reference: oracle march / hash / composite kernels, fp32 torch
    autograd for the two MLPs, loss = mean squared error against `target` with a white background."""
    n = o.shape[0]
    lv = oracle.make_levels(2**19, 16, 16, 1024, 2)
    hits = oracle.ray_aabb(o, d, 0.5)
    rays_a, xyzs, dirs, deltas, ts, S = oracle.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
    x01 = ((xyzs + 0.5) / 1.0).astype(np.float32)
    enc = torch.from_numpy(oracle.hash_fwd_f32(x01, table, lv)).requires_grad_(True)
    w = [torch.from_numpy(a.copy()).requires_grad_(True) for a in weights]
    dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    sh = torch.from_numpy(oracle.sh16_fwd(((dn + 1) / 2).astype(np.float32)))
    h = torch.relu(enc @ w[0].T) @ w[1].T
    sigma = torch.exp(h[:, 0])
    rgbs = torch.sigmoid(torch.relu(torch.relu(torch.cat([sh, h], 1) @ w[2].T) @ w[3].T) @ w[4].T)
    tot, op, dep, rgb, ws = oracle.composite_train_fwd(sigma.detach().numpy(), rgbs.detach().numpy(), deltas, ts, rays_a, 1e-4)
    rgb_f = rgb + (1.0 - op)[:, None]
    g_rgb = (2.0 / (3 * n) * (rgb_f - target)).astype(np.float32)
    g_op = (-g_rgb.sum(1)).astype(np.float32)
    ds, dc = oracle.composite_train_bwd(g_op, None, g_rgb, None, sigma.detach().numpy(), rgbs.detach().numpy(), deltas, ts, rays_a, 1e-4)
    torch.autograd.backward([sigma, rgbs], [torch.from_numpy(ds), torch.from_numpy(dc)])
    dtable = oracle.hash_bwd_f32(x01, enc.grad.numpy(), lv)
    return dtable, [t.grad.numpy() for t in w]


def test_end_to_end_gradients_vs_fp32_cpu_chain(oracle, conditioned):
    """FusedTrainer.compute_gradients (table + 5 weight gradients, fp16 MFMA MLP) against the fp32 CPU chain.  The yardstick is
    what torch's own autocast path (operator path: torch Linear layers under autocast, HIP hash / composite operators) loses
    against the same fp32 chain on the same rays."""
    from modules.rendering import render
    from ngp_hip import synthetic
    from ngp_hip.trainer import FusedTrainer
    n = 2048
    state = conditioned["state"]
    o, d = synthetic.lego_rays(n, seed=43)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    target = synthetic.procedural_render_gt(to, td).contiguous()
    # (1) HIP fused trainer
    m1 = _model("f32", state)
    tr = FusedTrainer(m1, init_scale=2.0**15)
    noise = _noise_for(n, 80)
    out = tr.compute_gradients(to, td, target, noise=torch.from_numpy(noise).cuda())
    g_table = out["table_grad"].cpu().numpy()
    g_mlp = np.split(out["mlp_grad"].cpu().numpy(), np.cumsum([2048, 1024, 2048, 4096])[:4])
    assert int(out["found_inf"]) == 0
    # (2) fp32 CPU chain
    weights = [w.detach().cpu().numpy() for w in m1._mlp_weights()]
    table = m1.pos_encoder.hash_table.detach().float().view(-1).cpu().numpy()
    r_table, r_mlp = _cpu_fp32_gradients(oracle, weights, table, o, d, conditioned["bits"], noise, target.cpu().numpy())
    # (3) torch autocast through the operator path (torch Linear layers), loss-scaled like the reference's GradScaler
    m3 = _model("f32", state)
    m3.use_fused_mlp = False
    _noise_for(n, 80)
    os.environ["NGP_FUSED_RENDER"] = "0"
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            res = render(m3, to, td, exp_step_factor=0.0)
            loss = torch.nn.functional.mse_loss(res["rgb"], target)
        (loss * 2.0**15).backward()
    finally:
        os.environ["NGP_FUSED_RENDER"] = "1"
    a_table = (m3.pos_encoder.hash_table.grad / 2.0**15).float().view(-1).cpu().numpy()
    a_mlp = [(w.grad / 2.0**15).float().cpu().numpy().reshape(-1) for w in m3._mlp_weights()]

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    names = ["table", "W1", "W2", "W3", "W4", "W5"]
    hip = [rel(g_table, r_table)] + [rel(g, r.reshape(-1)) for g, r in zip(g_mlp, r_mlp)]
    auto = [rel(a_table, r_table)] + [rel(a, r.reshape(-1)) for a, r in zip(a_mlp, r_mlp)]
    for nm, e1, e3 in zip(names, hip, auto):
        print("grad %-5s: HIP vs fp32 chain %.3e   torch-autocast vs fp32 chain %.3e" % (nm, e1, e3))
    assert np.linalg.norm(r_table) > 0 and all(np.linalg.norm(r) > 0 for r in r_mlp)
    # same support as the oracle's scatter-add (identical indexing); up to entries whose sum cancels / underflows
    touched_ref, touched_hip = r_table != 0, g_table != 0
    # (which sums cancel depends on the summation order, and the order of the live list / of the rays' ranges is the order blocks
    # finish in: 0.18-0.23 % over runs.  What has to hold: the entries in question carry nothing.)
    bad = touched_ref != touched_hip
    worst = max(float(np.abs(r_table[bad]).max(initial=0.0)), float(np.abs(g_table[bad]).max(initial=0.0)))
    print("support mismatch fraction %.5f, largest |value| there %.3e of max |grad| %.3e" % (bad.mean(), worst, np.abs(r_table).max()))
    assert bad.mean() < 4e-3 and worst <= 1e-4 * np.abs(r_table).max()      # (fp16 underflow of tiny output gradients: ~1e-5 of the largest)
    for nm, e1, e3 in zip(names, hip, auto):
        assert e1 <= max(1.5 * e3, 2e-3), (nm, e1, e3)      # no worse than torch's own fp16 autocast (plus a small floor)
        # ... and an ABSOLUTE bound beside the relative yard-stick: against the fp32 chain the fp16-autocast arithmetic
        # the reference itself runs loses 5e-4 .. 5e-3 on these gradients (both columns above, run after run: the fp16 forward, not the
        # backward, sets it); north_star's 1e-3 is stated for rendered radiance and PSNR (held in the tests above and in
        # test_gpu_trajectory.py), not for gradients
        assert e1 <= 8e-3, (nm, e1)
