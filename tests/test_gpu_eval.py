"""Evaluation render: the one-shot chunked path (march -> shade -> composite, modules/rendering.py::_render_rays_test_oneshot)
against the reference-shaped progressive loop (raymarching_test / composite_test rounds, rendering.py:62-158 of the reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(lego_bitfield, gain):
    from modules.networks import NGP
    torch.manual_seed(0)
    m = NGP(scale=0.5, max_res=1024).cuda().eval()
    m.density_bitfield.copy_(torch.from_numpy(lego_bitfield).cuda())
    with torch.no_grad():
        m.pos_encoder.hash_table.mul_(0.5)
        m.xyz_encoder.output_layer.weight.mul_(gain)       # gain > 1: part of the rays saturates and terminates early
    return m


@pytest.mark.parametrize("gain", [1.0, 400.0])
def test_oneshot_eval_matches_progressive_loop(hip_lib, lego_bitfield, monkeypatch, gain):
    from modules.rendering import render
    from ngp_hip import synthetic
    m = _model(lego_bitfield, gain)
    o, d = synthetic.lego_rays(30000, seed=21)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NGP_FUSED_EVAL", mode)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out[mode] = render(m, o, d, test_time=True, exp_step_factor=0.0)
    a, b = out["1"], out["0"]
    hit = b["opacity"] > 0
    assert hit.float().mean().item() > 0.1
    if gain > 1:
        assert (b["opacity"] > 0.999).float().mean().item() > 0.02            # saturated rays exist: early termination is exercised
    for k, tol in (("opacity", 2e-3), ("rgb", 2e-3), ("depth", 4e-3)):
        err = (a[k].float() - b[k].float()).abs().max().item()
        assert err < tol, (k, err)
    # chunking does not matter
    monkeypatch.setenv("NGP_FUSED_EVAL", "1")
    import modules.rendering as R
    monkeypatch.setattr(R, "EVAL_CHUNK", 7000)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        c = render(m, o, d, test_time=True, exp_step_factor=0.0)
    assert torch.equal(c["rgb"], a["rgb"]) and torch.equal(c["opacity"], a["opacity"]) and int(c["total_samples"]) == int(a["total_samples"])
