"""CPU restatement of the reference's optimisation loop (train.py:137-201) on top of the oracle kernels.

TEST INFRASTRUCTURE ONLY: imported by tests/ (tests/test_gpu_trajectory.py, tests/test_oracle_train_loop.py) -- never by anything
under taichi-nerfs_amd/.  It is the K-step yard-stick the HIP trainer's loss / PSNR trajectory is held to.

What one `step()` restates, statement by statement:
    render()                       modules/rendering.py:161-229   ray_aabb -> raymarching_train -> model(xyzs, dirs) -> VolumeRenderer
                                                                    -> background blend (white for exp_step_factor == 0, :219-226)
    model(x, d)                    modules/networks.py:136-166     x01 = (x - xyz_min) / (xyz_max - xyz_min); hash encode; xyz_encoder
                                                                    32 -> 64 ReLU -> 16; sigma = TruncExp(h[:, 0]); d / |d|; SH16((d+1)/2);
                                                                    rgb_net 32 -> 64 ReLU -> 64 ReLU -> 3 Sigmoid
    TruncExp                       modules/networks.py:18-30       exp forward, backward exponent clamped to [-15, 15]
    F.mse_loss(results['rgb'], data['rgb'])                         train.py:191
    + distortion_loss_w * distortion_loss(results).mean()           train.py:194-195, modules/distortion.py:8-119 (optional)
    Adam(eps=1e-15) + CosineAnnealingLR(max_steps, lr / 30)         train.py:143-160, :197-201  (torch's own CPU implementations)
    model.update_density_grid(thr, warmup=True)                     modules/networks.py:255-290 (all cells, :168-179; decay / max merge
                                                                    :281-284; threshold = min(mean positive density, thr) :286-290)
The march, hash encode (fp32 and half2), SH16, compositing forward / backward and packbits are the C oracle's (oracle/ngp_oracle.c,
each function citing the reference lines it follows); the two MLPs run in fp32 through torch-CPU autograd.  There is no GradScaler on
this side: in fp32 it is the identity unless a step overflows, and the tests run the GPU side at a loss scale that never does
(asserted there).  The half2 encoder's f16 rounding points are kept (f16 table copy per call, f16 embedding, f16 output gradient
scaled by `loss_scale` as hash_encoder_half.py:200-213 sees it under GradScaler).
"""
import math

import numpy as np
import torch

from . import ngp_oracle as ora


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


class OracleTrainer:
    """fp32 CPU training loop.  `weights`: the five [out, in] MLP matrices (networks.py order: xyz_encoder hidden, xyz_encoder output,
    rgb_net hidden 0, hidden 1, output); `table`: flat fp32 hash table; kind: 'f32' (hash_encoder.py) or 'half' (hash_encoder_half.py)."""

    def __init__(self, weights, table, scale=0.5, max_res=1024, exp_step_factor=0.0, lr=1e-2, max_steps=20000, kind="f32",
                 loss_scale=1.0, grid_size=128, max_samples=1024, T_threshold=1e-4, distortion_loss_w=0.0):
        assert kind in ("f32", "half")
        self.kind, self.scale, self.esf = kind, float(scale), float(exp_step_factor)
        self.cascades = max(1 + int(np.ceil(np.log2(2 * scale))), 1)                         # networks.py:63
        self.G, self.max_samples, self.T_threshold = int(grid_size), int(max_samples), float(T_threshold)
        self.bg = 1.0 if exp_step_factor == 0 else 0.0                                        # rendering.py:219-226
        self.loss_scale = float(loss_scale)
        self.distortion_loss_w = float(distortion_loss_w)                                     # train.py:194-195 (0 = off, the default)
        self.lv = ora.make_levels(2**19, 16, 16, max_res, 2)
        self.table = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32).reshape(-1).copy()).requires_grad_(True)
        self.w = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).copy()).requires_grad_(True) for a in weights]
        G3 = self.G**3
        self.density_grid = np.zeros((self.cascades, G3), np.float32)                         # networks.py:70-73
        self.bits = np.zeros(self.cascades * G3 // 8, np.uint8)
        self.opt = torch.optim.Adam([self.table] + self.w, lr, eps=1e-15)                     # train.py:151-155
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, max_steps, lr / 30)  # train.py:158-162
        self._cells = ora.morton3d_invert(np.arange(G3, dtype=np.int32))                      # cell i <-> Morton code i

    # ---------------------------------------------------------------------------------------------------- pieces
    def _encode(self, x01):
        t = self.table.detach().numpy()
        if self.kind == "half":                                                               # hash_encoder_half.py:367: cast per call
            return ora.hash_fwd_f16(x01, t.reshape(-1, 2).astype(np.float16), self.lv).reshape(-1, 32).astype(np.float32)
        return ora.hash_fwd_f32(x01, t, self.lv)

    def _x01(self, xyzs):
        s = np.float32(self.scale)
        return ((xyzs - (-s)) / (s - (-s))).astype(np.float32)                                # networks.py:144

    def density(self, xyzs_w, autocast=False):
        """networks.py:136-150 without gradient: sigma at world positions.  autocast: the two Linear layers as torch.autocast(fp16)
        evaluates them -- fp16 operands and results, fp32 accumulation -- with TruncExp in fp32 (its custom_fwd cast, networks.py:18-24)."""
        enc = self._encode(self._x01(xyzs_w))
        w1, w2 = self.w[0].detach().numpy(), self.w[1].detach().numpy()[:1]
        if autocast:
            h16 = lambda a: a.astype(np.float16).astype(np.float32)
            hid = h16(np.maximum(h16(h16(enc) @ h16(w1).T), 0))
            return np.exp(h16(hid @ h16(w2).T)[:, 0].astype(np.float32))
        with torch.no_grad():
            h0 = (torch.relu(torch.from_numpy(enc) @ self.w[0].T) @ self.w[1][:1].T)[:, 0]
            return torch.exp(h0).numpy()

    def update_density_grid(self, density_threshold, jitter, decay=0.95, autocast=True):
        """Warm-up form (all cells of every cascade, networks.py:168-179,255-290).  jitter[c]: [G^3, 3] uniforms in [0, 1), row i
        belonging to the cell with Morton code i.  autocast (default): train.py:177-182 calls the update INSIDE torch.autocast(fp16),
        so the reference's densities here come from fp16 Linear layers (the training forward of this loop stays fp32)."""
        G = self.G
        fresh = np.zeros_like(self.density_grid)                                              # :261
        cf = self._cells.astype(np.float32)
        for c in range(self.cascades):
            s = np.float32(min(2.0**(c - 1), self.scale))                                     # :270
            hg = np.float32(s / np.float32(G))                                                # :271
            base = (cf / np.float32(G - 1) * np.float32(2) - np.float32(1)) * np.float32(s - hg)           # :272-273
            xyzs_w = (base + (np.asarray(jitter[c], np.float32) * np.float32(2) - np.float32(1)) * hg).astype(np.float32)   # :275
            fresh[c] = self.density(xyzs_w, autocast)                                                 # :276 (indices = Morton code = row)
        g = self.density_grid
        self.density_grid = np.where(g < 0, g, np.maximum(g * np.float32(decay), fresh)).astype(np.float32)      # :281-284
        pos = self.density_grid[self.density_grid > 0]
        mean = float(pos.astype(np.float64).mean()) if pos.size else 0.0                      # :286
        self.bits = ora.packbits(self.density_grid.reshape(-1), min(mean, float(density_threshold)))         # :288-290
        return mean

    def forward_backward(self, o, d, target, noise, bits=None):
        """One render + loss + backward.  Leaves the gradients in .grad of the table / weights; returns the step's record."""
        n = o.shape[0]
        bits = self.bits if bits is None else bits
        hits = ora.ray_aabb(o, d, self.scale)
        rays_a, xyzs, dirs, deltas, ts, S = ora.march_train(o, d, hits, bits, noise, self.cascades, self.scale, self.esf, self.G,
                                                            self.max_samples)
        x01 = self._x01(xyzs)
        enc = torch.from_numpy(self._encode(x01)).requires_grad_(True)
        w = self.w
        dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)                               # networks.py:162
        sh = torch.from_numpy(ora.sh16_fwd(((dn + 1) / 2).astype(np.float32)))                # networks.py:163
        h = torch.relu(enc @ w[0].T) @ w[1].T
        sigma = _TruncExp.apply(h[:, 0])
        rgbs = torch.sigmoid(torch.relu(torch.relu(torch.cat([sh, h], 1) @ w[2].T) @ w[3].T) @ w[4].T)
        sg, cl = sigma.detach().numpy(), rgbs.detach().numpy()
        vr, op, dep, rgb, ws = ora.composite_train_fwd(sg, cl, deltas, ts, rays_a, self.T_threshold)
        rgb_f = rgb + np.float32(self.bg) * (1.0 - op)[:, None]                               # rendering.py:219-226
        diff = (rgb_f - target).astype(np.float32)
        loss = mse = float((diff.astype(np.float64)**2).mean())                               # train.py:191
        g_rgb = (2.0 / (3 * n) * diff).astype(np.float32)
        g_op = (-self.bg * g_rgb.sum(1)).astype(np.float32)
        g_ws = None
        if self.distortion_loss_w > 0:                                                        # loss += w * distortion_loss(results).mean()
            dl, ws_inc, wts_inc = ora.distortion_fwd(ws, deltas, ts, rays_a)                  # distortion.py:15-66
            loss += self.distortion_loss_w * float(dl.astype(np.float64).mean())
            g_dl = np.full(n, self.distortion_loss_w / n, np.float32)
            g_ws = ora.distortion_bwd(g_dl, deltas, ws, ts, ws_inc, wts_inc, rays_a)          # distortion.py:69-119
        ds, dc = ora.composite_train_bwd(g_op, None, g_rgb, g_ws, sg, cl, deltas, ts, rays_a, self.T_threshold)
        for p in [self.table] + self.w:
            p.grad = None
        if S > 0:
            torch.autograd.backward([sigma, rgbs], [torch.from_numpy(ds), torch.from_numpy(dc)])
            d_enc = enc.grad.numpy()
            if self.kind == "half":          # f16 output gradient under the loss scale, f16 products (hash_encoder_half.py:200-213)
                dt = ora.hash_bwd_f16(x01, (d_enc * np.float32(self.loss_scale)).astype(np.float16).reshape(-1, 16, 2), self.lv)
                dtable = (dt.reshape(-1) / np.float32(self.loss_scale)).astype(np.float32)
            else:
                dtable = ora.hash_bwd_f32(x01, d_enc, self.lv)
            self.table.grad = torch.from_numpy(dtable)
        else:
            self.table.grad = torch.zeros_like(self.table)
            for p in self.w:
                p.grad = torch.zeros_like(p)
        order = np.argsort(rays_a[:, 0], kind="stable")
        return {"loss": loss, "mse": mse, "psnr": -10.0 * math.log10(max(mse, 1e-30)), "rm_samples": int(S), "counts": rays_a[order, 2].copy(),
                "vr": vr.copy(), "rgb": rgb_f, "opacity": op}

    def step(self, o, d, target, noise, bits=None):
        rec = self.forward_backward(np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32),
                                    np.ascontiguousarray(target, np.float32), np.ascontiguousarray(noise, np.float32), bits)
        self.opt.step()                                                                       # train.py:199 (scaler.step == opt.step in fp32)
        self.sched.step()                                                                     # train.py:201
        return rec
