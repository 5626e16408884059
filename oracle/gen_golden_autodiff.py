"""Backward fixtures DERIVED FROM THE REFERENCE'S SOURCE: tests/golden/ref_*_grad.npz.

The three autodiff backwards of the hot path -- hash_encoder.py:269, spherical_harmonics.py:92, volume_train.py:160 -- exist
only inside Taichi's compiler.  Here the reference's own autograd glue (`HashEncoder`, `DirEncoder`, `VolumeRenderer` imported
from /root/reference/modules) is run end to end on CPU torch tensors under oracle/ti_shim, whose `kernel.grad(...)` re-executes
the reference's FORWARD kernel source under a reverse-mode float32 tape (ti_shim/taichi/_autodiff.py) with Taichi's torch-interop
conventions (`.grad` of the outputs = seeds, adjoints accumulated into `.grad` of the inputs).  Nothing of this repo's closed-form
backward code is involved: the oracle's closed forms and the HIP kernels are then held to these vectors.
    python oracle/gen_golden_autodiff.py          # build container only (needs /root/reference)
Also recorded: the factor 2 the reference's glue puts on a LEAF parameter's gradient (hash_encoder.py:277 returns `params.grad`,
which Taichi has already accumulated into the parameter's own .grad; autograd then adds it once more)."""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, golden_table, load_reference  # noqa: E402


def main():
    warnings.filterwarnings("ignore", message="The .grad attribute of a Tensor that is not a leaf")
    R = load_reference()
    rng = np.random.default_rng(404)
    T = torch.from_numpy

    # ------------------------------------------------------------ a-4  hash encoder (fp32), both level tables
    for tag, max_res in (("c2", 1024.0), ("c3", 4096.0)):
        enc = R["hash_encoder"].HashEncoder(max_params=2**19, levels=16, base_res=16.0, max_res=max_res, feature_per_level=2)
        table = golden_table(enc.total_param_size)
        n = 48
        x = rng.random((n, 3), dtype=np.float32)
        x[:6] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.25], [0.999999, 1e-7, 0.5], [0.3, 1.0, 0.0]]
        x[6:14] = x[14:22] + rng.normal(0, 2e-4, (8, 3)).astype(np.float32)      # near-duplicates: shared cells on every level
        x = np.clip(x, 0, 1).astype(np.float32)
        dout = rng.normal(0, 1, (n, 32)).astype(np.float32)
        dout[20:24] = 0.0
        # (1) the Taichi-level adjoint: kernel.grad with a non-leaf table -> d(table) exactly once
        tab = T(table.copy()).requires_grad_(True)
        params = tab * 1.0
        out = enc._module_function(T(x), params)
        out.backward(T(dout))
        g1 = tab.grad.numpy().copy()
        # (2) through the module exactly as networks.py uses it (leaf nn.Parameter): the glue's factor 2
        with torch.no_grad():
            enc.hash_table.copy_(T(table))
        enc.hash_table.grad = None
        enc(T(x)).backward(T(dout))
        g2 = enc.hash_table.grad.numpy().copy()
        rows = np.flatnonzero((g1.reshape(-1, 2) != 0).any(1))
        ratio = g2.reshape(-1, 2)[rows] / np.where(g1.reshape(-1, 2)[rows] == 0, 1, g1.reshape(-1, 2)[rows])
        assert np.all((g2.reshape(-1, 2)[rows] == 2 * g1.reshape(-1, 2)[rows])), "the module-level gradient is not exactly 2x"
        import taichi as ti_shim
        log_b = np.float32(enc.log_b)
        scale_used = np.array([np.float32(enc.base_res) * ti_shim.exp(np.float32(l) * log_b) - np.float32(1.0) for l in range(16)],
                              dtype=np.float32)
        np.savez_compressed(os.path.join(OUT, "ref_hash_f32_%s_grad.npz" % tag), xyzs=x, dout=dout, out=out.detach().numpy(),
                            grad_rows=rows.astype(np.int64), grad_vals=g1.reshape(-1, 2)[rows], module_leaf_factor=np.float32(ratio.max()),
                            scale_used=scale_used, total_entries=np.int64(enc.total_param_size // 2))
        print("hash %s grad ok: %d touched rows, leaf-parameter factor %.1f" % (tag, len(rows), ratio.max()))

    # ------------------------------------------------------------ a-6  SH16
    dirs = rng.random((96, 3), dtype=np.float32)
    dout = rng.normal(0, 1, (96, 16)).astype(np.float32)
    d = T(dirs.copy()).requires_grad_(True)
    R["spherical_harmonics"].DirEncoder()(d * 1.0).backward(T(dout))
    np.savez_compressed(os.path.join(OUT, "ref_sh16_grad.npz"), dirs=dirs, dout=dout, ddirs=d.grad.numpy())
    print("sh grad ok")

    # ------------------------------------------------------------ a-7  volume rendering (train)
    counts = np.array([0, 7, 33, 1, 120, 0, 12, 64], np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays_a = np.stack([np.array([5, 0, 2, 7, 1, 3, 6, 4], np.int32), starts, counts], -1)
    S = int(counts.sum())
    # one padding sample behind the last ray: the reference writes T[s + 1] past the end for a last ray that never terminates
    sig = (rng.random(S + 1, dtype=np.float32) * 30).astype(np.float32); sig[starts[4]:starts[4] + 120] *= 10
    rgbs = rng.random((S + 1, 3), dtype=np.float32); deltas = np.full(S + 1, 1.7320508 / 1024, np.float32)
    ts = (np.sort(rng.random(S + 1, dtype=np.float32)) + 0.5).astype(np.float32)
    g_op = rng.normal(0, 1, 8).astype(np.float32); g_dp = rng.normal(0, 1, 8).astype(np.float32)
    g_rgb = rng.normal(0, 1, (8, 3)).astype(np.float32); g_ws = rng.normal(0, 1, S + 1).astype(np.float32)
    s_ = T(sig.copy()).requires_grad_(True); c_ = T(rgbs.copy()).requires_grad_(True)
    tot, op, dp, rgb, ws = R["volume_train"].VolumeRenderer()(s_ * 1.0, c_ * 1.0, T(deltas), T(ts), T(rays_a), 1e-4)
    torch.autograd.backward([op, dp, rgb, ws], [T(g_op), T(g_dp), T(g_rgb), T(g_ws)])
    np.savez_compressed(os.path.join(OUT, "ref_composite_train_grad.npz"), sigmas=sig, rgbs=rgbs, deltas=deltas, ts=ts, rays_a=rays_a,
                        g_opacity=g_op, g_depth=g_dp, g_rgb=g_rgb, g_ws=g_ws, opacity=op.detach().numpy(), depth=dp.detach().numpy(),
                        rgb=rgb.detach().numpy(), d_sigmas=s_.grad.numpy(), d_rgbs=c_.grad.numpy(), n_valid=np.int64(S))
    print("composite train grad ok")


if __name__ == "__main__":
    main()
