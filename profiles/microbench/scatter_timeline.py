"""Per-level task timeline of the scatter-add launch INSIDE a real FusedTrainer step (ngp_hash_bwd_sliced_debug: 100 MHz stamps per task),
for either bench scene:  python profiles/microbench/scatter_timeline.py --scene lego|garden [--condition 512]
Prints per level: tasks, replicas, mean accumulate / flush time, when its tasks start and end, and per XCD the busy time against the span."""
import argparse
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
    ap.add_argument("--scene", default="lego", choices=["lego", "garden"])
    ap.add_argument("--condition", type=int, default=None)
    ap.add_argument("--dump", default=None)
    args = ap.parse_args()
    from ngp_hip import lib, synthetic
    from ngp_hip.ops import _ptr
    from ngp_hip.trainer import FusedTrainer
    from modules.networks import NGP
    lib.build(); L = lib.load()
    dev = torch.device("cuda")
    garden = args.scene == "garden"
    n = 65536 if garden else 8192
    cond = args.condition if args.condition is not None else (512 if garden else 1024)
    torch.manual_seed(23)
    model = NGP(scale=16.0 if garden else 0.5, max_res=4096 if garden else 1024).to(dev)
    tr = FusedTrainer(model, lr=1e-2, max_steps=20000, exp_step_factor=1 / 256 if garden else 0.0, distortion_loss_w=1e-3 if garden else 0.0)
    pool = []
    for b in range(8):
        o, d = (synthetic.garden_rays if garden else synthetic.lego_rays)(n, seed=1000 + 97 * b)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        tgt = synthetic.garden_render_gt(o, d, scale=16.0) if garden else synthetic.procedural_render_gt(o, d)
        pool.append((o, d, tgt.contiguous()))
    thr = 0.01 * 1024 / 3**0.5
    tr.set_deterministic(True)
    for i in range(cond):
        if i % 16 == 0:
            tr.update_density_grid(thr, warmup=i < 256)
        tr.step(*pool[i % 8])
    tr.set_deterministic(False)
    for i in range(cond, cond + 6):
        tr.step(*pool[i % 8])
    torch.cuda.synchronize()
    dbg = torch.zeros(8 * 1536, device=dev, dtype=torch.int64)
    L.ngp_hash_bwd_sliced_debug(_ptr(dbg))
    tr.step(*pool[(cond + 6) % 8])
    torch.cuda.synchronize()
    L.ngp_hash_bwd_sliced_debug(_ptr(None))
    live = int(tr._live_total[0])
    d = dbg.view(1536, 8).cpu().numpy()
    d = d[d[:, 1] > 0]
    t0 = d[:, 1].min()
    span = (d[:, 5].max() - t0) / 100.0
    print("%s: %d live samples, %d tasks, launch span %.1f us (100 MHz clock) = %.3f ns per live sample" % (args.scene, live, len(d), span, span * 1e3 / live))
    lv = model.pos_encoder.levels_struct
    for l in range(16):
        m = (d[:, 0] & 0xf) == l
        if not m.any():
            continue
        r = d[m]
        reps = len(set(((r[:, 0] >> 10) & 0x3f).tolist()))
        print("  level %2d res %5d entries %7d %s: %4d tasks (%2d replicas)  accumulate %7.1f us  flush %5.1f  start %7.1f..%7.1f  end<=%7.1f  CU-us %8.0f"
              % (l, lv.resolution[l], lv.map_size[l], "dense " if l < lv.begin_fast_hash_level else "hashed", m.sum(), reps, np.mean(r[:, 4] - r[:, 2]) / 100.0,
                 np.mean(r[:, 5] - r[:, 4]) / 100.0, (r[:, 1].min() - t0) / 100.0, (r[:, 1].max() - t0) / 100.0, (r[:, 5].max() - t0) / 100.0,
                 (r[:, 5] - r[:, 1]).sum() / 100.0))
    for x in range(8):
        r = d[d[:, 6] == x]
        if len(r):
            print("  xcc %d: %3d tasks, busy %8.1f CU-us (%.1f us over 32 CUs), last end %7.1f" % (x, len(r), (r[:, 5] - r[:, 1]).sum() / 100.0,
                                                                                                 (r[:, 5] - r[:, 1]).sum() / 100.0 / 32, (r[:, 5].max() - t0) / 100.0))
    if args.dump:
        np.save(args.dump, d)


if __name__ == "__main__":
    main()
