"""Scatter-add formulations on REAL backward inputs: condition a FusedTrainer on the analytic scene (like bench.py), keep the
last step's live list / positions / d_enc in the arena, then time ngp_hash_bwd_f32_live (float atomics) against
ngp_hash_bwd_f32_sliced (LDS-owned slices) and its plan knobs with HIP events, and compare the gradients.

    python profiles/microbench/hash_bwd_variants.py [--condition 1024] [--reps 20]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402



_KNOBS = {"bwd_knobs_dynamic": "1"}


def _knob(**kw):
    """(round 6) the plan / diagnostic knobs live behind NGP_EXPERIMENT; bwd_knobs_dynamic makes the library re-read them per call."""
    _KNOBS.update({k: str(v) for k, v in kw.items()})
    os.environ["NGP_EXPERIMENT"] = ";".join("%s=%s" % kv for kv in _KNOBS.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--condition", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rays", type=int, default=8192)
    args = ap.parse_args()
    _knob()                                                 # before the library's first plan: it latches bwd_knobs_dynamic then
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
    M = tr._march_sets(n)[1 - tr._cur]                      # the set the last step shaded
    cfg = RenderConfig(model, 0.0, 1e-4, 1024)
    live, total = int(tr._live_total[0]), int(M.total[0])
    if os.environ.get("NGP_VARIANTS_FRACTION"):             # only the first fraction of the live list: T(n) vs T(n / 2) gives the
        tr._live_total[0] = int(live * float(os.environ["NGP_VARIANTS_FRACTION"]))     # per-task cost that does not scale with hits
        live = int(tr._live_total[0])
    print("live samples %d, marched %d" % (live, total))
    lv = cfg.levels
    grad = torch.zeros_like(tr.table)
    ws = A.sliced_ws(lv)
    st = _stream()

    def atomic():
        return L.ngp_hash_bwd_f32_live(_ptr(M.xyzs), _ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), _ptr(A.live_idx), 1,
                                       cfg.lo, cfg.hi, 1, _ptr(grad), _ptr(None), st)

    def sliced():
        return L.ngp_hash_bwd_f32_sliced(_ptr(M.xyzs), _ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), _ptr(A.live_idx), 1,
                                         cfg.lo, cfg.hi, 1, _ptr(grad), _ptr(None), _ptr(ws), ws.numel(), st)

    def timeit(fn, reps):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            grad.zero_()
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return ts[len(ts) // 2], ts[0]

    out = {"live": live, "marched": total, "variants": {}}
    grad.zero_(); atomic(); ref = grad.clone()
    med, best = timeit(atomic, args.reps)
    out["variants"]["atomic (round 1)"] = {"median_us": med, "min_us": best}
    print("atomic: median %.1f us  min %.1f us" % (med, best))
    _knob(bwd_rep_target=48, bwd_merge_res=128, bwd_dense_min_rep=8, bwd_merge_chunks=32)      # the shipped plan
    # per-block timeline of one full launch
    dbg = torch.zeros(8 * 1536, device=dev, dtype=torch.int64)
    L.ngp_hash_bwd_sliced_debug(_ptr(dbg))
    grad.zero_(); sliced(); torch.cuda.synchronize()
    L.ngp_hash_bwd_sliced_debug(_ptr(None))
    d = dbg.view(1536, 8).cpu().numpy()
    d = d[d[:, 1] > 0]
    t0 = d[:, 1].min()
    print("timeline (us, 100 MHz clock): %d blocks, span %.1f" % (len(d), (d[:, 5].max() - t0) / 100.0))
    import numpy as np
    for lvl in range(16):
        m = (d[:, 0] & 0xf) == lvl
        if m.any():
            r = d[m]
            print("  level %2d: %3d tasks  start %6.1f..%6.1f  init %5.1f  accumulate %6.1f (wave0 %6.1f)  flush %5.1f  end<=%6.1f  xcc %s" % (
                lvl, m.sum(), (r[:, 1].min() - t0) / 100.0, (r[:, 1].max() - t0) / 100.0, np.mean(r[:, 2] - r[:, 1]) / 100.0,
                np.mean(r[:, 4] - r[:, 2]) / 100.0, np.mean(r[:, 3] - r[:, 2]) / 100.0, np.mean(r[:, 5] - r[:, 4]) / 100.0,
                (r[:, 5].max() - t0) / 100.0, sorted(set(r[:, 6].tolist()))))
    for x in range(8):
        r = d[d[:, 6] == x]
        if len(r):
            print("  xcc %d: %3d tasks, busy %7.1f CU-us (%.1f us if spread over 32 CUs), first start %5.1f, last end %6.1f" % (
                x, len(r), (r[:, 5] - r[:, 1]).sum() / 100.0, (r[:, 5] - r[:, 1]).sum() / 100.0 / 32, (r[:, 1].min() - t0) / 100.0,
                (r[:, 5].max() - t0) / 100.0))
    if os.environ.get("NGP_VARIANTS_DUMP"):                # raw rows: task word, begin, init, wave0, all waves, end (100 MHz ticks), xcc, n
        np.save(os.environ["NGP_VARIANTS_DUMP"], d)
    if os.environ.get("NGP_VARIANTS_ONLY_TIMELINE"):
        return
    # where the time goes: the same launch with pieces switched off (results are wrong with these flags: timing only)
    for diag, what in ((0, "full"), (1, "no LDS adds"), (2, "no gathers"), (3, "no adds, no gathers"), (4, "no accumulate"), (6, "no accumulate, no gathers")):
        _knob(bwd_diag=diag)
        med, best = timeit(sliced, 5)
        print("sliced, %-12s: median %.1f us" % (what, med))
    _knob(bwd_diag=0)
    for l in []:
        _knob(bwd_levels=hex(1 << l))
        med, best = timeit(sliced, 5)
        print("sliced level %2d only: median %.1f us" % (l, med))
    _knob(bwd_levels="0xffffffff")
    sweep = os.environ.get("NGP_VARIANTS_SWEEP")           # plan knobs: "rep=32,64;merge=64,128;dmin=2,4"
    knobs = {"rep": [48], "merge": [128], "dmin": [8], "mchunks": [32]}
    if sweep:
        for part in sweep.split(";"):
            k, v = part.split("=")
            knobs[k] = [int(x) for x in v.split(",")]
    for rep_t in knobs["rep"]:
        for merge in knobs["merge"]:
            for dmin in knobs["dmin"]:
                for mc in knobs["mchunks"]:
                    _knob(bwd_rep_target=rep_t, bwd_merge_res=merge, bwd_dense_min_rep=dmin, bwd_merge_chunks=mc)
                    grad.zero_()
                    name = "sliced rep_target=%d merge_res=%d dense_min_rep=%d merge_chunks=%d" % (rep_t, merge, dmin, mc)
                    if sliced() != 0:
                        print(name + ": plan not expressible")
                        continue
                    err = float((grad - ref).abs().max() / ref.abs().max())
                    med, best = timeit(sliced, args.reps)
                    out["variants"][name] = {"median_us": med, "min_us": best, "max_rel_err_vs_atomic": err}
                    print("%s: median %.1f us  min %.1f us   max|d|/max|ref| %.2e" % (name, med, best, err))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
