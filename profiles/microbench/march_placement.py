"""Where does the NEXT batch's march go once the table's optimizer rides in the scatter-add's flush (round 5)?

One conditioned FusedTrainer (bench.py's scene regime, deterministic conditioning), then every configuration -- position of the
prefetch hook in the step (trainer._prefetch_at: 0 start of the step, 1 after the hash gather, 2 after the MLP forward, 3 before
the scatter-add, 4 after it; -1 = not prefetched, the march in line), launch shape (ngp_march_train_fused_shaped: waves per block,
idle LDS per block) and side-stream priority -- runs `--steps` steps per round, the configurations ROUND-ROBIN over `--rounds`
rounds so that the slow drift of the model (live samples per step) hits all of them alike.  Reported per configuration: ms per
step, live samples per step, ns per live sample (the step costs ~0.9 us per 1000 live samples, so the last column is what to
compare).

usage: python profiles/microbench/march_placement.py [--configs "pos:waves:pad:prio;..."] [--steps 160] [--rounds 3]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

DEFAULT = ("-1:16:0:low;3:16:0:low;3:16:0:def;0:16:0:low;0:16:0:def;0:4:0:def;0:4:82944:def;0:8:82944:def;0:4:41472:def;"
           "1:4:82944:def;2:4:82944:def;0:4:82944:low;2:16:0:def;2:4:0:def")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--condition", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--configs", default=DEFAULT)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from ngp_hip import lib, synthetic
    from ngp_hip.lib import check
    from ngp_hip.trainer import FusedTrainer
    from modules.networks import NGP
    lib.build(); L = lib.load()
    dev = torch.device("cuda")
    torch.manual_seed(23)
    model = NGP(scale=0.5, max_res=1024).to(dev)
    tr = FusedTrainer(model, lr=1e-2, max_steps=20000)
    tr._adaptive_prefetch = False            # this script sets position / shape / stream itself
    pool = []
    for b in range(16):
        o, d = synthetic.lego_rays(args.rays, seed=1000 + 97 * b)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        pool.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
    thr = 0.01 * 1024 / 3**0.5
    # the two side streams a configuration picks from
    side_def = torch.cuda.Stream(device=dev)
    if tr._side_low is not None:
        side_low = tr._side_low
    else:
        h, lo, hi = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        check(L.ngp_stream_create_low_priority(ctypes.byref(h), ctypes.byref(lo), ctypes.byref(hi)), "ngp_stream_create_low_priority")
        side_low = torch.cuda.ExternalStream(h.value, device=dev)

    live_log = torch.zeros(4096, device=dev, dtype=torch.int32)

    def step(i, prefetch, k=None):
        if i % 16 == 0:
            tr.update_density_grid(thr, warmup=i < 256)
        nxt = pool[(i + 1) % 16]
        pre = (nxt[0], nxt[1]) if (prefetch and (i + 1) % 16 != 0) else None
        tr.step(*pool[i % 16], prefetch=pre)
        if k is not None:
            live_log[k].copy_(tr._live_total[0])

    tr.set_deterministic(True)
    for i in range(args.condition):
        step(i, True)
    tr.set_deterministic(False)
    torch.cuda.synchronize()
    configs = []
    for c in args.configs.split(";"):
        pos, waves, pad, prio = c.split(":")
        configs.append((int(pos), int(waves), int(pad), prio))
    res = {c: [] for c in configs}
    i = args.condition
    for rnd in range(args.rounds):
        for c in configs:
            pos, waves, pad, prio = c
            tr._prefetch_at = max(pos, 0)
            tr._march_shape = None if (waves == 16 and pad == 0) else (waves, pad)
            tr._side = side_low if prio == "low" else side_def
            for _ in range(8):
                step(i, pos >= 0); i += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(i, pos >= 0, k if k % 7 == 0 else None); i += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            live = float(live_log[0:args.steps:7].float().mean())
            res[c].append((dt / args.steps * 1e3, live))
    out = []
    print("%-28s %10s %12s %14s" % ("pos:waves:pad:prio", "ms/step", "live/step", "ns/live sample"))
    for c in configs:
        ms = sum(r[0] for r in res[c]) / len(res[c])
        live = sum(r[1] for r in res[c]) / len(res[c])
        per = [r[0] * 1e6 / r[1] for r in res[c]]
        print("%-28s %10.4f %12.0f %14.4f   rounds: %s" % ("%d:%d:%d:%s" % c, ms, live, ms * 1e6 / live, " ".join("%.4f" % x for x in per)))
        out.append({"config": "%d:%d:%d:%s" % c, "ms_per_step": ms, "live": live, "ns_per_live": ms * 1e6 / live, "rounds": per})
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
