"""A/B of the forward hash gather with and without paired x-edge loads (hash_grid.hip, hash_fwd_f32_xcd_kernel<0, PAIR>).
The switch (NGP_HASH_FWD_PAIR) is read once per process, so each variant runs in its own subprocess on identical inputs: samples
marched through the committed Lego occupancy (8192 rays x 4 batches ~ 400 k+ samples, sample order = the march's packing), a
seeded table.  Prints the median launch time (HIP events) per variant and whether the outputs are bit-identical.

    python profiles/microbench/hash_fwd_pair.py

Result: profiles/r04_hash_fwd_pair_experiment.txt (paired loads were slower; the PAIR variant and its switch were removed again, so
at HEAD both subprocesses time the same kernel -- the harness is kept for the next idea about this gather)."""
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
        ts = []
        for rep in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ngp_hash_fwd_f32_ex(_ptr(x), _ptr(table), lv, n, None, 1, -0.5, 0.5, 1, _ptr(enc), _stream())
            e1.record()
            assert rc == 0
            ts.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[5:])
        out[name] = {"n": n, "median_us": us[len(us) // 2], "min_us": us[0],
                     "sha256": hashlib.sha256(enc.cpu().numpy().tobytes()).hexdigest()}
    print("RESULT " + json.dumps(out))


def main():
    res = {}
    for rep in range(2):
        for v in ("0", "1"):
            env = dict(os.environ, NGP_HASH_FWD_V1=v)
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(o.stdout[-2000:], o.stderr[-2000:]); sys.exit(1)
            res.setdefault(v, []).append(json.loads(line[0][7:]))
    for cfg in res["0"][0]:
        a, b = [min(r[cfg]["median_us"] for r in res[v]) for v in ("0", "1")]
        same = len({r[cfg]["sha256"] for v in res for r in res[v]}) == 1
        print(f"{cfg:28s} n={res['0'][0][cfg]['n']}  one gather per corner {a:7.1f} us   paired x-edges {b:7.1f} us   ({b / a - 1:+.1%})   "
              f"outputs bit-identical: {same}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
