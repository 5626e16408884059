"""(The list-driven forms this script A/B-tests were measured slower and are not in the shipped library: experiment commits 3017399
(64 slices, one owner per CU) and 2df8868 (128 slices, two owners per CU).  At HEAD the script still times the prepass and the main
launch separately, prints the per-level task timeline and A/B-tests whatever knobs NGP_AB_VARIANTS names.)
Round 3 A/B of the scatter-add's hashed levels on REAL backward inputs: hit LISTS (NGP_BWD_LIST=1, the default) against round 2's
bitmap scan (NGP_BWD_LIST=0).  The plan knobs are read once per process, so run it once per setting:

    NGP_BWD_LIST=1 python profiles/microbench/hash_bwd_list_ab.py ; NGP_BWD_LIST=0 python profiles/microbench/hash_bwd_list_ab.py

Conditions a FusedTrainer on the analytic scene like bench.py, keeps the last step's live list / positions / d_enc, then times
the prepass (bitmaps [+ lists]) and the main launch separately with HIP events, prints the per-level task timeline of one launch
and compares the gradient with the float-atomic kernel's."""
import argparse
import ctypes
import json
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
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--out", default=None)
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
    M = tr._march_sets(n)[1 - tr._cur]                      # the set the last step shaded
    cfg = RenderConfig(model, 0.0, 1e-4, 1024)
    live, total = int(tr._live_total[0]), int(M.total[0])
    lv = cfg.levels
    lm = ctypes.c_uint32(0)
    if hasattr(L, "ngp_hash_bwd_sliced_list_plan"):                # (the list owners live in commits 3017399 / 2df8868, not at HEAD)
        L.ngp_hash_bwd_sliced_list_plan(ctypes.byref(lv), _ptr(None), 0, _ptr(None), _ptr(None), ctypes.byref(lm))
    list_levels = lm.value
    print("live samples %d, marched %d, list-driven levels 0x%04x (NGP_BWD_LIST=%s)" % (live, total, list_levels, os.environ.get("NGP_BWD_LIST", "unset")))
    grad = torch.zeros_like(tr.table)
    ws = A.sliced_ws(lv)
    st = _stream()

    def atomic():
        return L.ngp_hash_bwd_f32_live(_ptr(M.xyzs), _ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), _ptr(A.live_idx), 1,
                                       cfg.lo, cfg.hi, 1, _ptr(grad), _ptr(None), st)

    def prep():
        return L.ngp_hash_bwd_sliced_prep(_ptr(M.xyzs), ctypes.byref(lv), A.cap, _ptr(tr._live_total), _ptr(A.live_idx), 1, cfg.lo, cfg.hi,
                                          _ptr(ws), ws.numel(), st)

    def main_():
        return L.ngp_hash_bwd_sliced_main(_ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(tr._live_total), 1, _ptr(grad), _ptr(None), _ptr(ws),
                                          ws.numel(), st)

    def timeit(fn, reps, zero=False):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            if zero:
                grad.zero_()
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return ts[len(ts) // 2], ts[0]

    out = {"live": live, "marched": total, "list_levels": list_levels, "NGP_BWD_LIST": os.environ.get("NGP_BWD_LIST")}
    grad.zero_(); assert atomic() == 0; ref = grad.clone()
    grad.zero_(); assert prep() == 0 and main_() == 0
    torch.cuda.synchronize()
    err = float((grad - ref).abs().max() / ref.abs().max())
    same_support = bool(torch.equal(grad != 0, ref != 0))
    out["max_rel_err_vs_atomic"], out["same_support"] = err, same_support
    print("vs float-atomic kernel: max|d|/max|ref| %.2e, same touched entries: %s" % (err, same_support))
    g2, r2 = grad.view(-1, 2), ref.view(-1, 2)
    for lvl in range(16):
        a, b = int(lv.offset[lvl]), int(lv.offset[lvl]) + int(lv.map_size[lvl])
        ga, ra = g2[a:b], r2[a:b]
        sup = int(((ga != 0) != (ra != 0)).any(1).sum())
        if sup or os.environ.get("NGP_AB_VERBOSE"):
            dd = (ga - ra).abs()
            only_g = int(((ga != 0).any(1) & ~(ra != 0).any(1)).sum()); only_r = int((~(ga != 0).any(1) & (ra != 0).any(1)).sum())
            print("  level %2d: %d rows differ in support (only sliced %d, only atomic %d), max|d| %.3e, max|ref| %.3e, sum sliced %.6e sum ref %.6e" % (
                lvl, sup, only_g, only_r, float(dd.max()), float(ra.abs().max()), float(ga.double().sum()), float(ra.double().sum())))
    if list_levels and os.environ.get("NGP_AB_CHECK_LISTS", "0") == "1":        # (written for the 64-slice lists of commit 3017399)
        # integrity of the hit lists of one level: every live sample exactly once per (y, z) combination, filed under the right slice
        CH = 2048
        ms = (A.cap + CH - 1) // CH * CH
        off_pool = ms * 12 + 16 * 64 * (ms // 64) * 8
        n_list = bin(list_levels).count("1")
        pool_level, tab_level = 4 * ms, (ms // CH) * 64
        pool_t = ws[off_pool:off_pool + n_list * pool_level * 4].view(torch.int32).view(n_list, -1)
        tab_t = ws[off_pool + n_list * pool_level * 4:off_pool + n_list * (pool_level + tab_level) * 4].view(torch.int32).view(n_list, -1, 64)
        n_chunks = (live + CH - 1) // CH
        xyzc = ws[:ms * 12].view(torch.float32).view(-1, 3)[:live]
        for lo, lvl in enumerate([l for l in range(16) if (list_levels >> l) & 1]):
            tab_l = tab_t[lo, :n_chunks].long() & 0xffffffff
            start, cnt = tab_l & 0xffff, tab_l >> 16
            tot = cnt.sum(1)
            expect = torch.full((n_chunks,), 4 * CH, device=dev); expect[-1] = 4 * (live - (n_chunks - 1) * CH)
            ok_counts = bool(torch.equal(tot, expect))
            ok_starts = bool(torch.equal(start, torch.cumsum(cnt, 1) - cnt))
            ent = pool_t[lo, :n_chunks * 4 * CH].view(n_chunks, 4 * CH).long() & 0xffffffff
            valid = torch.arange(4 * CH, device=dev)[None, :] < tot[:, None]
            i_s, k_s = (ent & 0x3fffffff)[valid], (ent >> 30)[valid]
            key = i_s * 4 + k_s
            uniq = torch.unique(key)
            ok_perm = bool(uniq.numel() == 4 * live and int(key.max()) == 4 * live - 1 and key.numel() == 4 * live)
            # the slice each entry was filed under vs the one its corner pair hashes to
            pos_in_chunk = torch.arange(4 * CH, device=dev)[None, :].expand(n_chunks, -1)
            bounds = (start + cnt)                                                   # [n_chunks, 64] exclusive ends
            filed = (pos_in_chunk[:, :, None] >= bounds[:, None, :]).sum(-1)[valid]
            sc = float(lv.scale[lvl])
            cy = torch.floor(xyzc[i_s, 1] * sc + 0.5).long(); cz = torch.floor(xyzc[i_s, 2] * sc + 0.5).long()
            Ah = (((cy + (k_s & 1)) * 2654435761) ^ ((cz + (k_s >> 1)) * 805459861)) & (int(lv.map_size[lvl]) - 1)
            ok_slice = bool(torch.equal(Ah >> 13, filed))
            print("  lists level %2d: counts %s starts %s permutation %s slices %s" % (lvl, ok_counts, ok_starts, ok_perm, ok_slice))
    out["prep_us"] = timeit(prep, args.reps)
    out["main_us"] = timeit(main_, args.reps, zero=True)
    print("prepass: median %.1f us (min %.1f)   main: median %.1f us (min %.1f)" % (out["prep_us"] + out["main_us"]))
    if os.environ.get("NGP_BWD_KNOBS_DYNAMIC"):
        # same process, same inputs: alternate the two formulations
        out["ab"] = []
        variants = [v.split(",") for v in os.environ.get("NGP_AB_VARIANTS", "NGP_BWD_LIST=1;NGP_BWD_LIST=0").split(";")]
        keys = sorted({kv.split("=")[0] for v in variants for kv in v})
        saved = {k: os.environ.get(k) for k in keys}
        for rnd in range(3):
            for v in variants:
                for kv in v:
                    k, val = kv.split("=")
                    os.environ[k] = val
                grad.zero_(); assert prep() == 0 and main_() == 0
                e_ = float((grad - ref).abs().max() / ref.abs().max())
                pm, mm = timeit(prep, args.reps), timeit(main_, args.reps, zero=True)
                out["ab"].append({"variant": ",".join(v), "prep_us": pm[0], "main_us": mm[0], "err": e_})
                print("  A/B round %d %-32s: prepass %.1f us  main %.1f us  sum %.1f   err %.1e" % (rnd, ",".join(v), pm[0], mm[0], pm[0] + mm[0], e_))
                if os.environ.get("NGP_AB_TIMELINE") and rnd == 0:
                    dbg_ = torch.zeros(8 * 3072, device=dev, dtype=torch.int64)
                    L.ngp_hash_bwd_sliced_debug(_ptr(dbg_)); grad.zero_(); main_(); torch.cuda.synchronize(); L.ngp_hash_bwd_sliced_debug(_ptr(None))
                    d_ = dbg_.view(3072, 8).cpu().numpy(); d_ = d_[d_[:, 1] > 0]
                    print("      tasks %d span %.1f | mean task us per level: %s" % (len(d_), (d_[:, 5].max() - d_[:, 1].min()) / 100.0, " ".join(
                        "%d:%.1f" % (l_, np.mean((d_[(d_[:, 0] & 0xf) == l_][:, 5] - d_[(d_[:, 0] & 0xf) == l_][:, 1])) / 100.0) for l_ in range(16) if ((d_[:, 0] & 0xf) == l_).any())))
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        prep()                                              # the workspace again holds this mode's lists / bitmaps for the timeline below
    dbg = torch.zeros(8 * 3072, device=dev, dtype=torch.int64)
    L.ngp_hash_bwd_sliced_debug(_ptr(dbg))
    grad.zero_(); main_(); torch.cuda.synchronize()
    L.ngp_hash_bwd_sliced_debug(_ptr(None))
    d = dbg.view(3072, 8).cpu().numpy()
    d = d[d[:, 1] > 0]
    t0 = d[:, 1].min()
    span = (d[:, 5].max() - t0) / 100.0
    busy = (d[:, 5] - d[:, 1]).sum() / 100.0
    print("timeline (us, 100 MHz clock): %d tasks, span %.1f, task time summed %.0f CU-us = %.1f us on 256 CUs" % (len(d), span, busy, busy / 256))
    out["timeline"] = {"tasks": int(len(d)), "span_us": float(span), "busy_cu_us": float(busy), "levels": {}}
    for lvl in range(16):
        m = (d[:, 0] & 0xf) == lvl
        if m.any():
            r = d[m]
            row = {"tasks": int(m.sum()), "init": float(np.mean(r[:, 2] - r[:, 1]) / 100.0), "accumulate": float(np.mean(r[:, 4] - r[:, 2]) / 100.0),
                   "flush": float(np.mean(r[:, 5] - r[:, 4]) / 100.0), "task": float(np.mean(r[:, 5] - r[:, 1]) / 100.0),
                   "last_end": float((r[:, 5].max() - t0) / 100.0)}
            out["timeline"]["levels"][lvl] = row
            print("  level %2d: %3d tasks  init %4.1f  accumulate %5.1f  flush %4.1f  task %5.1f  last end %6.1f  xcc %s" % (
                lvl, row["tasks"], row["init"], row["accumulate"], row["flush"], row["task"], row["last_end"], sorted(set(r[:, 6].tolist()))))
    print(json.dumps(out))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
