"""A/B harness for the two position-driven encoder launches: the forward hash gather (ngp_hash_fwd_f32_ex, pair-major output) and
the scatter-add's prepass (ngp_hash_bwd_sliced_prep).  Library switches are read once per process, so every variant runs in its
own subprocess on identical inputs: samples marched through the committed Lego occupancy (8192 rays x 4 batches, ~750 k samples
in the march's own packing order) and a seeded table.  Prints the median launch time (HIP events) per variant, and whether the
gather's output / the prepass's workspace are bit-identical across variants.

    python profiles/microbench/encoder_ab.py NGP_HASH_FWD_V1 0 1        # round-4 gather loop vs the round-1..3 loop
    python profiles/microbench/encoder_ab.py NGP_PREP_BATCH 1 3 6 12    # LDS-form levels per fence in the prepass

Results: profiles/r04_hash_fwd_loop_experiment.txt, profiles/r04_hash_fwd_pair_experiment.txt (the paired-load variant of that
file is not in the tree)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    from ngp_hip import lib as L, ops, synthetic
    from ngp_hip.ops import _ptr, _stream
    lib = L.load()
    bits = torch.from_numpy(np.load(os.path.join(ROOT, "tests/golden/lego_density_bitfield.npz"))["density_bitfield"]).cuda()
    xs = []
    for seed in range(4):
        o, d = [torch.from_numpy(a).cuda() for a in synthetic.lego_rays(8192, seed=seed)]
        hits = ops.ray_aabb(o, d, 0.5)
        g = torch.Generator(device="cuda").manual_seed(seed)
        noise = torch.rand(8192, device="cuda", generator=g)
        rays_a, xyzs, dirs, deltas, t_mid, total = ops.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
        xs.append(xyzs[: int(total)])
    x = torch.cat(xs).contiguous()
    n = x.shape[0]
    out = {}
    for name, args in (("C2 T=2^19", (2 ** 19, 16, 16, 1024)), ("C3 T=2^21 max_res 4096", (2 ** 21, 16, 16, 4096))):
        lv = ops.make_levels(args[0], args[1], args[2], args[3], 2)
        torch.manual_seed(1)
        table = (torch.rand(lv.total_entries * 2, device="cuda") * 2 - 1) * 1e-1
        enc = torch.empty(8 * n * 4, device="cuda")
        ws = ops.sliced_workspace(lv, n, torch.device("cuda", 0))
        ws.zero_()
        ts, tp = [], []
        for rep in range(25):
            e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e0.record()
            rc = lib.ngp_hash_fwd_f32_ex(_ptr(x), _ptr(table), lv, n, None, 1, -0.5, 0.5, 1, _ptr(enc), _stream())
            e1.record()
            rc2 = lib.ngp_hash_bwd_sliced_prep(_ptr(x), lv, n, None, None, 1, -0.5, 0.5, _ptr(ws), ws.numel(), _stream())
            e2.record()
            assert rc == 0 and rc2 in (0, -2), (rc, rc2)         # -2: more than 64 slices per level (C3): no LDS-sliced scatter-add
            ts.append((e0, e1)); tp.append((e1, e2))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[5:])
        up = sorted(a.elapsed_time(b) * 1e3 for a, b in tp[5:])
        out[name] = {"n": n, "fwd_us": us[len(us) // 2], "prep_us": up[len(up) // 2],
                     "fwd_sha": hashlib.sha256(enc.cpu().numpy().tobytes()).hexdigest()[:16],
                     "prep_sha": hashlib.sha256(ws.cpu().numpy().tobytes()).hexdigest()[:16]}
    print("RESULT " + json.dumps(out))


def main():
    args = [a for a in sys.argv[1:] if a != "--child"]
    var, values = args[0], args[1:]
    res = {}
    for rep in range(2):
        for v in values:
            env = dict(os.environ, **{var: v})
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(o.stdout[-2000:], o.stderr[-2000:]); sys.exit(1)
            res.setdefault(v, []).append(json.loads(line[0][7:]))
    for cfg in res[values[0]][0]:
        print(f"{cfg}  (n = {res[values[0]][0][cfg]['n']})")
        for v in values:
            f, q = [min(r[cfg][k] for r in res[v]) for k in ("fwd_us", "prep_us")]
            print(f"   {var}={v:3s}  gather {f:7.1f} us   prepass {q:7.1f} us")
        for k in ("fwd_sha", "prep_sha"):
            print(f"   {k[:-4]} results bit-identical across variants and repeats: {len({r[cfg][k] for v in res for r in res[v]}) == 1}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
