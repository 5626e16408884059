"""The fused MLP kernels on REAL step inputs: condition a FusedTrainer on the analytic scene (like bench.py), keep the last step's
arena (encodings, directions, output gradients, live list) and time ngp_mlp_fwd_ex / ngp_mlp_bwd_live alone with HIP events.
Variants are chosen per call through the environment (NGP_MLP_* knobs the library reads at launch), so A/B runs share one process
and identical inputs:
    NGP_AB_VARIANTS="NGP_MLP_BWD_BLOCKS=256;NGP_MLP_BWD_BLOCKS=512" python profiles/microbench/mlp_time.py [--condition 512]
"""
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
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rays", type=int, default=8192)
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
    live, total = int(tr._live_total[0]), int(M.total[0])
    print("live samples %d, marched %d" % (live, total))
    st = _stream()
    P = tr.enc_pairs
    found = ctypes.c_void_p(tr.state_i.data_ptr())
    d_enc = torch.empty_like(A.d_enc)
    dW = torch.zeros_like(tr.mlp_grad)
    sig = torch.empty_like(A.sigmas); rgb = torch.empty_like(A.rgbs)

    def bwd():
        return L.ngp_mlp_bwd_live(_ptr(A.enc), _ptr(M.dirs), _ptr(tr.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap, _ptr(tr._live_total),
                                  _ptr(A.live_idx), P, _ptr(d_enc), _ptr(dW), _ptr(None), st)

    def bwd_no_dw():                  # dW = NULL: the same launch without the end-of-kernel weight-gradient flush (2.6 M atomics)
        return L.ngp_mlp_bwd_live(_ptr(A.enc), _ptr(M.dirs), _ptr(tr.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap, _ptr(tr._live_total),
                                  _ptr(A.live_idx), P, _ptr(d_enc), _ptr(None), _ptr(None), st)

    parts = torch.empty(L.ngp_mlp_dw_parts_max() * 9408, device=dev)

    def bwd_parts():                  # per-block slabs instead of atomics (what the trainer launches; the sum rides in its prologue)
        rc = L.ngp_mlp_bwd_live_parts(_ptr(A.enc), _ptr(M.dirs), _ptr(tr.wpack), _ptr(A.d_sigmas), _ptr(A.d_rgbs), A.cap,
                                      _ptr(tr._live_total), _ptr(A.live_idx), P, _ptr(d_enc), _ptr(parts), _ptr(None), st)
        return 0 if rc > 0 else -1

    def reduce_():
        return L.ngp_mlp_dw_reduce(_ptr(parts), 256, _ptr(dW), st)

    def fwd():
        return L.ngp_mlp_fwd_ex(_ptr(A.enc), _ptr(M.dirs), _ptr(tr.wpack), A.cap, _ptr(M.total), P, _ptr(sig), _ptr(rgb), st)

    def timeit(fn, reps):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return ts[len(ts) // 2], ts[0]

    if hasattr(L, "ngp_mlp_bwd_occupancy"):
        print("mlp_bwd blocks per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor): %d" % L.ngp_mlp_bwd_occupancy())
    dW.zero_(); bwd(); torch.cuda.synchronize()
    ref_denc, ref_dW = d_enc.clone(), dW.clone()
    variants = [v for v in os.environ.get("NGP_AB_VARIANTS", "A=0").split(";") if v]
    for rnd in range(3):
        for v in variants:
            for kv in v.split(","):
                k, val = kv.split("=")
                os.environ[k] = val
            dW.zero_(); d_enc.zero_(); bwd(); torch.cuda.synchronize()
            e1 = float((d_enc - ref_denc).abs().max() / ref_denc.abs().max())
            e2 = float((dW - ref_dW).abs().max() / ref_dW.abs().max())
            mb, _ = timeit(bwd, args.reps)
            mb0, _ = timeit(bwd_no_dw, args.reps)
            mbp, _ = timeit(bwd_parts, args.reps)
            mred, _ = timeit(reduce_, args.reps)
            mf, _ = timeit(fwd, args.reps)
            if hasattr(L, "ngp_mlp_debug_read") and rnd == 0:        # -DNGP_MLP_DIAG build: where a round's cycles go (block 0)
                import numpy as np
                buf = np.zeros(128, np.uint64)
                L.ngp_mlp_debug_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
                bwd(); torch.cuda.synchronize()
                L.ngp_mlp_debug_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
                if os.environ.get("NGP_MLP_BWD", "reg")[0] != "l":           # register form: absolute stamps per wave of blocks 0..3
                    print("   register form, ticks since kernel entry per wave (blocks 0-3 x 4 waves): after weights | loop end | reduced | exit")
                    for w_, row in enumerate(buf.reshape(16, 8)):
                        print("     wave %2d: %8d %8d %8d %8d" % (w_, row[0], row[1], row[2], row[3]))
                    print("round %d %-40s: mlp_bwd %.1f us" % (rnd, v, mb))
                    continue
                seg = ["wait inputs", "forward", "dX chain", "phase A", "phase B", "phase C"]
                b = buf.reshape(16, 8)[:12, :6].astype(np.float64)
                print("   cycles per launch in block 0, mean over waves 0-7 / 8-11 (s_memtime ticks):")
                for k, name in enumerate(seg):
                    print("     %-12s %9.0f %9.0f" % (name, b[:8, k].mean(), b[8:, k].mean()))
                print("     total        %9.0f %9.0f" % (b[:8].sum(1).mean(), b[8:].sum(1).mean()))
            print("round %d %-40s: mlp_bwd %.1f us (%.1f without the dW flush, %.1f with slabs + %.1f stand-alone reduce)  mlp_fwd %.1f us   d_enc err %.1e  dW err %.1e" % (rnd, v, mb, mb0, mbp, mred, mf, e1, e2))


if __name__ == "__main__":
    main()
