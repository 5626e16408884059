"""What the level-group split of the scatter-add costs on ONE GPU ("1-GPU step unchanged within 2 % with the split
launch"): ngp_hash_bwd_sliced_main (one launch) against ngp_hash_bwd_sliced_main_levels issued once per level group, on the real
backward inputs of a conditioned FusedTrainer (same harness as hash_bwd_list_ab.py).  Groups are given as first levels in launch
order, e.g. "8,0" = levels 8-15 then 0-7; the later launches are also timed on a reduced number of persistent workgroups
(--blocks, what the overlapped exchange uses so that RCCL finds free CUs)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--condition", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--groups", default="8,0;12,8,0;12,0;10,0")
    ap.add_argument("--blocks", type=int, default=240)
    args = ap.parse_args()
    from ngp_hip import lib, synthetic
    from ngp_hip.fused import RenderConfig, TrainArena
    from ngp_hip.ops import _ptr, _stream
    from ngp_hip.trainer import FusedTrainer
    from modules.networks import NGP
    lib.build(); L = lib.load()
    dev = torch.device("cuda")
    torch.manual_seed(23)
    model = NGP(scale=0.5, max_res=1024).to(dev)
    tr = FusedTrainer(model, lr=1e-2, max_steps=20000)
    pool = []
    for b in range(16):
        o, d = synthetic.lego_rays(args.rays, seed=1000 + 97 * b)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        pool.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
    for i in range(args.condition):
        if i % 16 == 0:
            tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=i < 256)
        tr.step(*pool[i % 16])
    torch.cuda.synchronize()
    n = args.rays
    A = TrainArena.get(dev, n, 1024)
    M = tr._march_sets(n)[1 - tr._cur]
    cfg = RenderConfig(model, 0.0, 1e-4, 1024)
    lv, P, st = cfg.levels, tr.enc_pairs, _stream()
    live = int(tr._live_total[0])
    print("live samples %d" % live)
    grad = torch.zeros_like(tr.table)
    ws = A.sliced_ws(lv)
    assert L.ngp_hash_bwd_sliced_prep(_ptr(M.xyzs), ctypes.byref(lv), A.cap, _ptr(tr._live_total), _ptr(A.live_idx), 1, cfg.lo, cfg.hi,
                                      _ptr(ws), ws.numel(), st) == 0

    def whole():
        return L.ngp_hash_bwd_sliced_main(_ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), P, _ptr(grad), _ptr(None), _ptr(ws),
                                          ws.numel(), st)

    def split(starts, blocks):
        def run():
            hi = 16
            for k, l0 in enumerate(starts):
                mask = sum(1 << l for l in range(l0, hi))
                rc = L.ngp_hash_bwd_sliced_main_levels(_ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), P, _ptr(grad), _ptr(None),
                                                       _ptr(ws), ws.numel(), mask, 0 if k == 0 else blocks, st)
                if rc:
                    return rc
                hi = l0
            return 0
        return run

    def timeit(fn):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for e0, e1 in ev:
            grad.zero_()
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return ts[len(ts) // 2]

    grad.zero_(); whole(); ref = grad.clone()
    for rnd in range(2):
        print("round %d: one launch %.1f us" % (rnd, timeit(whole)))
        for spec in args.groups.split(";"):
            starts = [int(x) for x in spec.split(",")]
            grad.zero_(); split(starts, 0)(); torch.cuda.synchronize()
            same = bool(torch.equal(grad != 0, ref != 0)) and float((grad - ref).abs().max() / ref.abs().max()) < 1e-6
            print("   groups %-8s: %.1f us on 256 workgroups, %.1f us with the later launches on %d   (same gradient: %s)" % (
                spec, timeit(split(starts, 0)), timeit(split(starts, args.blocks)), args.blocks, same))


if __name__ == "__main__":
    main()
