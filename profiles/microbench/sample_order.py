"""Round 4 gate experiment: does a spatially coherent SAMPLE / RAY order shorten the scatter-add
(prepass + main), the MLP backward and the hash gather?  Nothing in the library changes: the harness permutes the inputs.

Part A -- live-list order.  A FusedTrainer is conditioned like bench.py; the last step's live list is permuted offline by a
sort key (sort time excluded), then `ngp_mlp_bwd_live` (writes d_enc at list positions), `ngp_hash_bwd_sliced_prep` and
`ngp_hash_bwd_sliced_main` are timed with HIP events on that list; the gradient is checked against the unpermuted run.
The forward gather is timed on a position array laid out in the same order.
Part B -- ray order.  The same conditioned trainer runs whole steps (march in line, no prefetch) with every batch's rays
permuted by a key before the march (per-ray results do not depend on the ray order); all kernels of the step are bracketed.
Variants alternate (A B C A B C ...) so the slow drift of the training state hits all of them alike."""
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


def part1by1(v):
    v = v & 0xffff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def part1by2(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton2(a, b):
    return part1by1(a.long()) | (part1by1(b.long()) << 1)


def morton3(a, b, c):
    return part1by2(a.long()) | (part1by2(b.long()) << 1) | (part1by2(c.long()) << 2)


class TimedLib:
    """Proxy of the ctypes library: the named entry points are bracketed with HIP events on torch's current stream."""

    def __init__(self, L, names):
        self._L, self._names, self.on, self.rec = L, set(names), False, {}

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        if name not in self._names:
            return fn

        def timed(*a):
            if not self.on:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*a); e1.record()
            self.rec.setdefault(name, []).append((e0, e1))
            return rc
        return timed

    def drain(self):
        torch.cuda.synchronize()
        out = {k: float(np.median([a.elapsed_time(b) * 1e3 for a, b in v])) for k, v in self.rec.items()}
        self.rec = {}
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--condition", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=24, help="part B: measured steps per variant")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only-order", default=None, help="part A on this one ordering only and no part B (for a rocprofv3 --pmc pass)")
    args = ap.parse_args()
    from ngp_hip import lib, ops, synthetic
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
    P = tr.enc_pairs
    st = _stream()
    out = {"live": live, "marched": total, "part_a": {}, "part_b": {}}
    print("live %d marched %d" % (live, total), flush=True)

    # ------------------------------------------------------------------ part A: live-list order
    base = A.live_idx[:live].clone()
    grad = torch.zeros_like(tr.table)
    dW = torch.zeros_like(tr.mlp_grad)
    ws = A.sliced_ws(lv)
    found = torch.zeros(1, device=dev, dtype=torch.int32)
    cnt = tr._live_total

    def mlp_bwd():
        return L.ngp_mlp_bwd_live(_ptr(A.enc), _ptr(M.dirs), _ptr(tr.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap, _ptr(cnt),
                                  _ptr(A.live_idx), P, _ptr(A.d_enc), _ptr(dW), _ptr(found), st)

    def prep():
        return L.ngp_hash_bwd_sliced_prep(_ptr(M.xyzs), ctypes.byref(lv), A.cap, _ptr(cnt), _ptr(A.live_idx), 1, cfg.lo, cfg.hi,
                                          _ptr(ws), ws.numel(), st)

    def main_():
        return L.ngp_hash_bwd_sliced_main(_ptr(A.d_enc), ctypes.byref(lv), A.cap, _ptr(cnt), P, _ptr(grad), _ptr(found), _ptr(ws),
                                          ws.numel(), st)

    xyz_scratch = torch.empty(live, 3, device=dev)
    enc_scratch = torch.empty(8 * live * 4 + 64, device=dev)
    n_dev = torch.tensor([live], device=dev, dtype=torch.int32)

    def gather():
        return L.ngp_hash_fwd_f32_ex(_ptr(xyz_scratch), _ptr(tr.table), ctypes.byref(lv), live, _ptr(n_dev), 1, cfg.lo, cfg.hi, P,
                                     _ptr(enc_scratch), st)

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
        return ts[len(ts) // 2]

    x = M.xyzs[base.long()]                                     # [live, 3] world positions
    xn = ((x - cfg.lo) / (cfg.hi - cfg.lo)).clamp(0, 1)

    def cell(level):
        return torch.floor(xn * float(lv.scale[level]) + 0.5).int()

    def key_yz(level):
        c = cell(level)
        return morton2(c[:, 1], c[:, 2])

    def key_xyz(level):
        c = cell(level)
        return morton3(c[:, 0], c[:, 1], c[:, 2])

    g128 = (xn * 128).int().clamp(0, 127)
    orders = {
        "as_shipped(block-completion)": None,
        "random": torch.randperm(live, device=dev),
        "yz_morton_L10": torch.argsort(key_yz(10), stable=True),
        "yz_morton_L13": torch.argsort(key_yz(13), stable=True),
        "yz_morton_L15": torch.argsort(key_yz(15), stable=True),
        "xyz_morton_128": torch.argsort(morton3(g128[:, 0], g128[:, 1], g128[:, 2]), stable=True),
        "xyz_morton_L10": torch.argsort(key_xyz(10), stable=True),
        "xyz_morton_L15": torch.argsort(key_xyz(15), stable=True),
        "zyx_lex_L15": torch.argsort((cell(15)[:, 2].long() << 22) | (cell(15)[:, 1].long() << 11) | cell(15)[:, 0].long(), stable=True),
    }
    ref = None
    if args.only_order:
        orders = {k: v for k, v in orders.items() if k.startswith(args.only_order)}
    for name, perm in orders.items():
        idx = base if perm is None else base[perm]
        A.live_idx[:live].copy_(idx)
        xyz_scratch.copy_(M.xyzs[idx.long()])
        grad.zero_(); dW.zero_()
        assert mlp_bwd() == 0 and prep() == 0 and main_() == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = grad.clone()
            err = 0.0
        else:
            err = float((grad - ref).abs().max() / ref.abs().max())
        row = {"mlp_bwd_us": timeit(mlp_bwd, args.reps), "prep_us": timeit(prep, args.reps), "main_us": timeit(main_, args.reps, zero=True),
               "gather_us": timeit(gather, args.reps), "err_vs_shipped": err}
        row["prep_plus_main_us"] = row["prep_us"] + row["main_us"]
        out["part_a"][name] = row
        print("A %-30s mlp_bwd %6.1f  prep %5.1f  main %6.1f  prep+main %6.1f  gather %5.1f   err %.1e" % (
            name, row["mlp_bwd_us"], row["prep_us"], row["main_us"], row["prep_plus_main_us"], row["gather_us"], err), flush=True)
    A.live_idx[:live].copy_(base)
    if args.only_order:
        print(json.dumps(out))
        return

    # ------------------------------------------------------------------ part B: ray order
    names = ["ngp_march_train_fused", "ngp_hash_fwd_f32_ex", "ngp_mlp_fwd_ex", "ngp_composite_train_fused_live", "ngp_hash_bwd_sliced_prep",
             "ngp_mlp_bwd_live", "ngp_hash_bwd_sliced_main", "ngp_adam_all_ex"]
    T = TimedLib(L, names)
    tr.L = T

    def ray_keys(o, d):
        # slab test in torch (harness only): entry / exit of the [-0.5, 0.5]^3 box
        inv = 1.0 / d
        ta, tb = (-0.5 - o) * inv, (0.5 - o) * inv
        tn = torch.minimum(ta, tb).amax(1).clamp(min=0)
        tf = torch.maximum(ta, tb).amin(1)
        hit = tf > tn
        pe = ((o + tn[:, None] * d + 0.5).clamp(0, 1) * 63.999).int()
        pm = ((o + (0.5 * (tn + tf))[:, None] * d + 0.5).clamp(0, 1) * 63.999).int()
        dn = d / d.norm(dim=1, keepdim=True)
        dq = ((dn * 0.5 + 0.5) * 63.999).int()
        big = (~hit).long() << 40
        return {
            "entry_morton64": morton3(pe[:, 0], pe[:, 1], pe[:, 2]) + big,
            "mid_morton64": morton3(pm[:, 0], pm[:, 1], pm[:, 2]) + big,
            "dir_then_entry": (morton3(dq[:, 0] >> 2, dq[:, 1] >> 2, dq[:, 2] >> 2) << 18) + morton3(pe[:, 0], pe[:, 1], pe[:, 2]) + big,
        }

    variants = ["as_shipped(random rays)", "entry_morton64", "mid_morton64", "dir_then_entry"]
    pools = {variants[0]: pool}
    for v in variants[1:]:
        pl = []
        for o, d, tg in pool:
            p = torch.argsort(ray_keys(o, d)[v], stable=True)
            pl.append((o[p].contiguous(), d[p].contiguous(), tg[p].contiguous()))
        pools[v] = pl
    acc = {v: {} for v in variants}
    lives = {v: [] for v in variants}
    wall = {v: [] for v in variants}
    step_no = args.condition
    for rnd in range(args.steps):
        for v in variants:
            o, d, tg = pools[v][rnd % 16]
            T.on = True
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); tr.step(o, d, tg); e1.record()
            T.on = False
            r = T.drain()
            for k, us in r.items():
                acc[v].setdefault(k, []).append(us)
            wall[v].append(e0.elapsed_time(e1) * 1e3)
            lives[v].append(int(tr._live_pair.max()))
        step_no += 1
    for v in variants:
        row = {k: float(np.median(us)) for k, us in acc[v].items()}
        row["step_us_with_events"] = float(np.median(wall[v]))
        row["kernel_sum_us"] = float(sum(row[k] for k in names if k in row))
        row["live"] = float(np.mean(lives[v]))
        out["part_b"][v] = row
        print("B %-26s live %7.0f | march %5.1f gather %5.1f mlp_fwd %5.1f comp %5.1f prep %5.1f mlp_bwd %5.1f main %6.1f adam %5.1f | sum %6.1f" % (
            v, row["live"], row.get(names[0], 0), row.get(names[1], 0), row.get(names[2], 0), row.get(names[3], 0), row.get(names[4], 0),
            row.get(names[5], 0), row.get(names[6], 0), row.get(names[7], 0), row["kernel_sum_us"]), flush=True)
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
