"""Where the one-launch march's time goes, per wave (-DNGP_MARCH_DIAG build): condition a FusedTrainer on the analytic scene like
bench.py, march one batch with the model's own occupancy grid, and read back every wave's (start, march done, block done) stamps.
    NGP_HIPCC_EXTRA=-DNGP_MARCH_DIAG python profiles/microbench/march_waves.py [--condition 1024]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--condition", type=int, default=1024)
    args = ap.parse_args()
    from ngp_hip import lib, ops, synthetic
    from ngp_hip.trainer import FusedTrainer
    from modules.networks import NGP
    lib.build(); L = lib.load()
    dev = torch.device("cuda")
    torch.manual_seed(23)
    model = NGP(scale=0.5, max_res=1024).to(dev)
    tr = FusedTrainer(model, lr=1e-2, max_steps=20000)
    pool = []
    for b in range(16):
        o, d = synthetic.lego_rays(8192, seed=1000 + 97 * b)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        pool.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
    for i in range(args.condition):
        if i % 16 == 0:
            tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=i < 256)
        tr.step(*pool[i % 16])
    torch.cuda.synchronize()
    o, d, _ = pool[3]
    noise = torch.rand(8192, device=dev)
    for _ in range(3):
        r = ops.march_train_fused(o, d, None, model.density_bitfield, noise, 1, 0.5, 0.0, 128, 1024)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ops.march_train_fused(o, d, None, model.density_bitfield, noise, 1, 0.5, 0.0, 128, 1024); e1.record()
    torch.cuda.synchronize()
    print("march launch %.1f us, %d samples (%.1f per ray)" % (e0.elapsed_time(e1) * 1e3, int(r[5]), int(r[5]) / 8192))
    buf = np.zeros(4 * 4096, np.uint64)
    assert L.ngp_march_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    b = buf.reshape(4096, 4).astype(np.float64)
    t0 = b[:, 0].min()
    start, done, end = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, (b[:, 2] - t0) / 100.0
    life = done - start
    print("waves: start %.1f..%.1f us; march of a wave's two rays: mean %.1f, median %.1f, p90 %.1f, p99 %.1f, max %.1f us; last block done at %.1f us"
          % (start.min(), start.max(), life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), end.max()))
    blk = done.reshape(256, 16)
    print("per block: slowest wave done at mean %.1f us (min %.1f, max %.1f); mean over ALL waves %.1f us -> a perfectly balanced launch "
          "would be ~%.1f us + expansion" % (blk.max(1).mean(), blk.max(1).min(), blk.max(1).max(), done.mean(), life.mean()))
    hist, edges = np.histogram(life, bins=[0, 5, 10, 15, 20, 25, 30, 40, 50, 80])
    print("lifetime histogram (us):", {"%d-%d" % (edges[k], edges[k + 1]): int(hist[k]) for k in range(len(hist))})


if __name__ == "__main__":
    main()
