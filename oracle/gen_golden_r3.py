"""Round-3 fixtures, generated like oracle/gen_golden.py by EXECUTING THE REFERENCE'S OWN SOURCE from /root/reference (Taichi
kernels under oracle/ti_shim, torch code as it is).  Separate script so that the round-1/2 fixtures keep reproducing bit for bit.
Run in the build container only:   python oracle/gen_golden_r3.py

  ref_hash_f16_big.npz        a-5: the half2 encoder (modules/hash_encoder_half.py:112-213) on 256 points incl. box faces / corners;
                              forward output, explicit backward, and for every touched table row the number of SAMPLES that touch
                              it (rows with one owner do not depend on the accumulation order: they must match bit for bit)
  ref_update_density_grid.npz f-3: NGP.update_density_grid / sample_uniform_and_occupied_cells / get_all_cells
                              (modules/networks.py:168-209,255-290) on a 2-cascade 16^3 grid with an analytic density field, the
                              random draws recorded; warm-up and sampled update, decay/max merge, invisible cells, bitfield
  ref_mark_invisible.npz      NGP.mark_invisible_cells (modules/networks.py:212-253) on the same grid, 5 cameras
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def analytic_density(x):
    """x [n,3] world positions -> sigma [n]: a blob + ripples; spans 0 .. ~50 so that the 0.01 * 1024 / sqrt(3) threshold cuts it."""
    r2 = (x * x).sum(-1)
    return 50.0 * torch.exp(-6.0 * r2) * (1.0 + 0.5 * torch.sin(9.0 * x[:, 0]) * torch.cos(7.0 * x[:, 1])) * (x[:, 2] > -0.3)


def hash_f16_big(R):
    H = R["hash_encoder_half"]
    T = torch.from_numpy
    rng = np.random.default_rng(2323)
    enc = H.HashEncoder(max_params=2**19, levels=16, base_res=16.0, max_res=1024.0, feature_per_level=2)
    n = 256
    x = rng.random((n, 3), dtype=np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.25], [0.999999, 1e-7, 0.5], [0.3, 1.0, 0.0], [0.25, 0.75, 1.0], [1.0, 0.5, 1.0]]
    x[8:40] = x[40:72] + rng.random((32, 3), dtype=np.float32) * 1e-3          # near-duplicates: shared cells on the fine levels too
    table_h = (gg.golden_table(enc.total_param_size, -0.1, 0.1)).astype(np.float16).reshape(-1, 2)
    out = torch.zeros(n, 16, 2, dtype=torch.float16)
    enc._hash_encoder_kernel(T(x), T(table_h), out, enc.hash_map_sizes, enc.offsets, n)
    dout = (rng.standard_normal((n, 16, 2)) * 1e-2).astype(np.float16)
    dout[::4] = 0
    grad = torch.zeros(table_h.shape[0], 2, dtype=torch.float16)
    enc._hash_encoder_backward_kernel(T(x), enc.hash_map_sizes, enc.offsets, T(dout), grad, n)
    owners = np.zeros(table_h.shape[0], np.int32)                              # samples touching each row (any contribution)
    for i in range(n):
        gi = torch.zeros(table_h.shape[0], 2, dtype=torch.float16)
        enc._hash_encoder_backward_kernel(T(x[i:i + 1]), enc.hash_map_sizes, enc.offsets, T(dout[i:i + 1]), gi, 1)
        owners += (np.abs(gi.numpy().astype(np.float32)).sum(1) != 0)
    nz = np.flatnonzero(owners)
    import taichi as ti_shim
    scale_used = np.array([np.float32(16.0) * ti_shim.exp(lvl * np.float32(enc.log_b)) - np.float32(1.0) for lvl in range(16)], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_hash_f16_big.npz"), xyzs=x, out=out.numpy(), dout=dout, grad_rows=nz, scale_used=scale_used,
                        grad_vals=grad.numpy()[nz], grad_owners=owners[nz], total_entries=np.int64(table_h.shape[0]))
    print("hash f16 big ok: %d touched rows, %d with a single owner" % (len(nz), int((owners[nz] == 1).sum())))


class Host:
    """Plain carrier of the attributes NGP's occupancy methods read (scale, cascades, grid_size, the three buffers, density())."""


def make_host(N, G, scale, cascades, seed):
    from kornia.utils.grid import create_meshgrid3d
    g = torch.Generator().manual_seed(seed)
    h = Host()
    h.scale, h.cascades, h.grid_size = scale, cascades, G
    G3 = G**3
    grid = torch.where(torch.rand(cascades, G3, generator=g) < 0.3, torch.rand(cascades, G3, generator=g) * 30.0, torch.zeros(cascades, G3))
    grid[0, ::7] = -1.0
    grid[1, 5::11] = -1.0
    h.density_grid = grid
    h.density_bitfield = torch.zeros(cascades * G3 // 8, dtype=torch.uint8)
    h.grid_coords = create_meshgrid3d(G, G, G, False, dtype=torch.int32).reshape(-1, 3)
    h.density = analytic_density
    for name in ("get_all_cells", "sample_uniform_and_occupied_cells", "update_density_grid", "mark_invisible_cells"):
        setattr(h, name, types.MethodType(getattr(N.NGP, name), h))
    return h


class Recorder:
    """Records (and optionally overrides) the draws of torch.randint / torch.rand_like inside the reference's functions."""

    def __init__(self, jitter_value=None):
        self.randint, self.rand_like, self.jitter_value = [], [], jitter_value
        self._ri, self._rl = torch.randint, torch.rand_like

    def __enter__(self):
        def randint(*a, **k):
            out = self._ri(*a, **k)
            self.randint.append(out.clone())
            return out

        def rand_like(x, *a, **k):
            out = self._rl(x, *a, **k) if self.jitter_value is None else torch.full_like(x, self.jitter_value)
            self.rand_like.append(out.clone())
            return out
        torch.randint, torch.rand_like = randint, rand_like
        return self

    def __exit__(self, *exc):
        torch.randint, torch.rand_like = self._ri, self._rl


def occupancy(N):
    G, scale, cascades = 16, 1.0, 2
    thr = 0.01 * 1024 / 3**0.5                                                  # train.py:180
    save = {"grid_size": np.int64(G), "scale": np.float64(scale), "cascades": np.int64(cascades), "threshold": np.float64(thr),
            "decay": np.float64(0.95)}
    # ---- warm-up update (all cells), random jitter recorded
    h = make_host(N, G, scale, cascades, seed=7)
    save["grid_before"] = h.density_grid.numpy().copy()
    torch.manual_seed(11)
    cells = h.get_all_cells()
    save["all_indices"] = cells[0][0].numpy().astype(np.int32)                  # Morton code of grid_coords row r
    with Recorder() as rec:
        h.update_density_grid(thr, warmup=True)
    save["warm_jitter"] = np.stack([j.numpy() for j in rec.rand_like])          # [cascades, G^3, 3] in grid_coords order
    save["warm_grid_after"] = h.density_grid.numpy().copy()
    save["warm_bitfield"] = h.density_bitfield.numpy().copy()
    # sigma per cell exactly as the reference evaluated it (the product's density network is not what is under test here)
    sig = np.zeros((cascades, G**3), np.float32)
    pos = np.zeros((cascades, G**3, 3), np.float32)
    for c in range(cascades):
        idx, coords = cells[c]
        s = min(2**(c - 1), scale)
        hg = s / G
        xyzs_w = (coords / (G - 1) * 2 - 1) * (s - hg)
        xyzs_w += (rec.rand_like[c] * 2 - 1) * hg
        sig[c, idx.numpy()] = analytic_density(xyzs_w).numpy()
        pos[c, idx.numpy()] = xyzs_w.numpy()
    save["warm_sigma_by_morton"] = sig
    save["warm_xyz_by_morton"] = pos
    # ---- sampled update from that state: M uniform + M occupied cells per cascade, jitter pinned to the cell centre (0.5) so
    # that a cell drawn twice has one density; the randint draws are recorded
    save["samp_grid_before"] = h.density_grid.numpy().copy()
    torch.manual_seed(12)
    with Recorder(jitter_value=0.5) as rec:
        h.update_density_grid(thr, warmup=False)
    M = G**3 // 4
    save["samp_coords1"] = np.stack([rec.randint[2 * c].numpy() for c in range(cascades)])          # [cascades, M, 3]
    save["samp_rand_idx"] = np.stack([rec.randint[2 * c + 1].numpy() for c in range(cascades)])     # [cascades, M] into nonzero(grid > thr)
    save["samp_n_occupied"] = np.array([(save["samp_grid_before"][c] > thr).sum() for c in range(cascades)], np.int64)
    save["samp_grid_after"] = h.density_grid.numpy().copy()
    save["samp_bitfield"] = h.density_bitfield.numpy().copy()
    sigc = np.zeros((cascades, G**3), np.float32)                               # density at every cell CENTRE (what any sampled cell gets)
    for c in range(cascades):
        idx, coords = cells[c]
        s = min(2**(c - 1), scale)
        hg = s / G
        xyzs_w = (coords / (G - 1) * 2 - 1) * (s - hg)
        xyzs_w += (torch.full_like(xyzs_w, 0.5) * 2 - 1) * hg
        sigc[c, idx.numpy()] = analytic_density(xyzs_w).numpy()
    save["centre_sigma_by_morton"] = sigc
    assert M == save["samp_coords1"].shape[1]
    np.savez_compressed(os.path.join(OUT, "ref_update_density_grid.npz"), **save)
    print("update_density_grid ok: occupied after warm-up %.3f, after sampled %.3f" % (
        np.unpackbits(save["warm_bitfield"]).mean(), np.unpackbits(save["samp_bitfield"]).mean()))

    # ---- mark_invisible_cells
    h = make_host(N, G, scale, cascades, seed=8)
    h.density_grid = torch.zeros(cascades, G**3)
    rng = np.random.default_rng(5)
    n_cam = 5
    pos_c = rng.standard_normal((n_cam, 3)); pos_c = 1.6 * pos_c / np.linalg.norm(pos_c, axis=1, keepdims=True)
    pos_c[4] = [0.05, 0.02, 0.2]                                                # one camera inside the grid: near-plane rejections
    fwd = -pos_c / np.linalg.norm(pos_c, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    poses = torch.tensor(np.concatenate([np.stack([right, down, fwd], -1), pos_c[..., None]], -1), dtype=torch.float32)
    K = torch.tensor([[150.0, 0, 40.0], [0, 150.0, 30.0], [0, 0, 1]])     # narrow field of view: a good part of the grid is seen by no camera
    img_wh = (80, 60)
    h.mark_invisible_cells(K, poses, img_wh, chunk=1000)
    np.savez_compressed(os.path.join(OUT, "ref_mark_invisible.npz"), K=K.numpy(), poses=poses.numpy(), img_wh=np.array(img_wh),
                        grid_size=np.int64(G), scale=np.float64(scale), cascades=np.int64(cascades),
                        density_grid=h.density_grid.numpy(), count_grid=h.count_grid.numpy())
    print("mark_invisible ok: %.3f of the cells invisible" % float((h.density_grid < 0).float().mean()))


def main():
    R = gg.load_reference()
    sys.path.append(os.path.join(ROOT, "taichi-nerfs_amd", "compat"))         # kornia stand-in (create_meshgrid3d) only; AFTER the shim
    N = importlib.import_module("refmodules.networks")
    os.makedirs(OUT, exist_ok=True)
    hash_f16_big(R)
    occupancy(N)


if __name__ == "__main__":
    main()
    gg.write_provenance()
