"""bench.py -- training rays/s of the Instant-NGP hot path on N MI355X (BASELINE.json metric, config C2/C4).

One "step" = one full optimisation step over one batch of 8192 synthetic Lego-shape rays per GPU:
ray-AABB -> occupancy march -> hash-grid encode -> MLPs (+SH) -> composite -> MSE -> backward (composite, MLPs,
hash scatter-add) -> (N>1: RCCL all-reduce of all gradients) -> GradScaler unscale + Adam; plus the reference's
occupancy-grid update every 16 steps (train.py:178-182), whose cost stays inside the timed region.

The measured state is a TRAINING state, not an initialisation artefact (default --regime scene): target colours are the
dense-integration render of an analytic Lego-shape scene (ngp_hip/synthetic.py), and before the warm-up the model is
conditioned, untimed, by --condition seeded optimisation steps on that scene with the reference's schedule (occupancy
warm-up for the first 256 steps, update every 16).  From there on the occupancy grid is the model's own, samples per ray
and the live fraction are stable, and `--steps 20 --warmup 5` measures the same step as `--steps 200 --warmup 20`.

Usage: python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL -- either launched by torch.distributed.run (WORLD_SIZE / RANK / LOCAL_RANK in the environment),
or, when WORLD_SIZE is not set, bench.py starts the N ranks itself (and fails loudly if the node has fewer than N GPUs); the
line then carries comm_ms / exposed_comm_ms / comm_breakdown_ms / rccl_version / rccl_env.
Prints ONE JSON line on rank 0.  At N = 1 the default line also carries `configs`: short runs of the other BASELINE configs.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA
# algorithmic bytes / FLOPs per unit of work (fp32 table, L=16, F=2)
BYTES_PER_SAMPLE = {"hash_fwd_f32": 12 + 1024 + 128, "hash_bwd_f32": 12 + 128 + 1024 + 1024}
# kernel (C-ABI entry) -> (key, bound, work per unit, unit, which argument is the unit count)
TRAINER_KERNELS = {
    "ngp_march_train_count_ex": ("march_count", "hbm", 24 + 8 + 4 + 12, "ray"),      # + 8 B per staged sample, added below
    "ngp_march_train_fused": ("march_count", "hbm", 24 + 8 + 4 + 12, "ray"),         # one-launch march: + (8 + 32) B per sample, added below
    "ngp_hash_fwd_f32_ex": ("hash_fwd_f32", "hbm", 12 + 1024 + 128, "sample"),
    "ngp_hash_fwd_f32": ("hash_fwd_f32", "hbm", 12 + 1024 + 128, "n_arg"),       # occupancy-update encodes (exact n = arg 3)
    "ngp_mlp_fwd_ex": ("mlp_fwd", "mfma", 18816, "sample"),
    # round 5, chunked forward (multi-cascade scenes): one launch per round over a list of samples; units = the step's shaded samples / rounds
    "ngp_hash_fwd_list": ("hash_fwd_f32", "hbm", 12 + 1024 + 128, "shaded"),
    "ngp_mlp_fwd_list": ("mlp_fwd", "mfma", 18816, "shaded"),
    "ngp_mlp_bwd_ex": ("mlp_bwd", "mfma", 37632, "sample"),
    "ngp_hash_bwd_f32_ex": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "sample"),
    "ngp_mlp_bwd_live": ("mlp_bwd", "mfma", 37632, "live"),                        # backward kernels run on the live-sample list
    "ngp_mlp_bwd_live_parts": ("mlp_bwd", "mfma", 37632, "live"),                  # ... weight gradients as per-block slabs (the trainer)
    "ngp_hash_bwd_f32_live": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),
    "ngp_hash_bwd_f32_sliced": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),   # LDS-sliced form (prep + main launch)
    "ngp_hash_bwd_sliced_main": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),  # ... its main launch (what the trainer issues)
    "ngp_hash_bwd_sliced_main_slabs": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),  # ... + the sum of the MLP backward's slabs at its head (round 4)
    # round 5: ... with the table's optimizer in its flush (the hashed levels never leave as a gradient): priced below as the scatter-add's
    # algorithmic bytes + the bytes the optimizer part really moves (12 B per parameter read, 12 B per touched parameter written)
    "ngp_hash_bwd_sliced_main_adam": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),
    "ngp_hash_bwd_sliced_main_adam_step": ("hash_bwd_f32", "hbm", 12 + 128 + 1024 + 1024, "live"),   # ... + the step's scalar bookkeeping
    "ngp_hash_bwd_sliced_prep": ("hash_bwd_prep", "hbm", 12 + 12 + 16 * 8, "live"),        # ... its prepass (in line, before the MLP backward)
    "ngp_hash_bwd_f16_live": ("hash_bwd_f16", "hbm", 12 + 128 + 512 + 512, "live"),
    "ngp_hash_bwd_sliced_main_f16": ("hash_bwd_f16", "hbm", 12 + 128 + 512 + 512, "live"),    # half2 encoder, LDS-sliced form
    "ngp_adam_step": ("adam", "hbm", 32, "param"),
    "ngp_adam_all": ("adam", "hbm", 32, "param"),                                  # table pass (+ the MLP block riding along)
    "ngp_hash_fwd_f16_ex": ("hash_fwd_f16", "hbm", 12 + 512 + 128, "sample"),      # --half: 4-byte gathers, f32 output to the arena
    "ngp_hash_bwd_f16_ex": ("hash_bwd_f16", "hbm", 12 + 128 + 512 + 512, "sample"),
    "ngp_adam_all_ex": ("adam", "hbm", 32, "param"),
    "ngp_hash_fwd_bf16_ex": ("hash_fwd_bf16", "hbm", 12 + 512 + 128, "sample"),    # --table bf16: 4-byte gathers, f32 output
    "ngp_adam_step_bf16": ("adam_bf16", "hbm", 34, "param"),                       # + the 2-byte storage copy
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU per step (default: 8192 for lego = BASELINE C2, 65536 for garden = C3)")
    ap.add_argument("--scene", default="lego", choices=["lego", "garden"],
                    help="lego = BASELINE C2 (scale 0.5, 1 cascade, max_res 1024, the headline); garden = BASELINE C3 shape "
                         "(scale 16, 6 cascades, max_res 4096, exponential stepping, black background, distortion loss 1e-3)")
    ap.add_argument("--regime", default="scene", choices=["scene", "lego", "random50", "ones"],
                    help="scene (default): analytic-scene targets + a conditioned model marching its OWN occupancy grid.  The others "
                         "are round-1's synthetic-state diagnostics (random target colours, the bitfield overwritten every update): "
                         "lego = trained-Lego fixture bitfield, random50 = seeded 50%% (initialisation regime), ones = all occupied")
    ap.add_argument("--condition", type=int, default=1024,
                    help="--regime scene: untimed, seeded optimisation steps on the scene before the warm-up (multiple of 16)")
    ap.add_argument("--no-deterministic-condition", dest="det_condition", action="store_false",
                    help="--regime scene, trainer path: by default the untimed conditioning steps run in the trainer's deterministic mode "
                         "(ray-order packing, one owner per table slice, no float atomics: two processes end conditioning in the same "
                         "state); warm-up and timed steps always run the default fast path")
    ap.add_argument("--pool", type=int, default=32, help="--regime scene: resident batches the steps cycle through")
    ap.add_argument("--step-trace", default=None, help="diagnostic: write per-step host/device times of the timed region to FILE")
    ap.add_argument("--kernel-events-every", type=int, default=4,
                    help="HIP events around the dominant kernel are recorded on every E-th timed step only (default 4, at most steps / 3; 1 = every step).  Each "
                         "event is a barrier packet between two kernels: on every step they cost 50-70 us per step (10 %% of it, "
                         "A/B in profiles/r02_bench_kernel_events_ab.txt); the per-kernel averages are the same either way")
    ap.add_argument("--no-kernel-events", dest="kernel_events", action="store_false",
                    help="no per-kernel HIP events inside the timed region (A/B for the event overhead)")
    ap.add_argument("--half", action="store_true", help="half2 hash encoder (BASELINE C5)")
    ap.add_argument("--table", default="f32", choices=["f32", "bf16"],
                    help="hash-table storage the forward gathers from: f32 (the reference's default encoder; the headline line) "
                         "or a bf16 copy of the fp32 master table (BASELINE config 2 wording)")
    ap.add_argument("--path", default="trainer", choices=["trainer", "modules"],
                    help="trainer: ngp_hip.trainer.FusedTrainer (device-resident step); modules: the reference's loop shape "
                         "(render() through modules/ + torch Adam + torch GradScaler)")
    ap.add_argument("--graph", action="store_true", help="replay the trainer step from a captured hipGraph (1 GPU)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="do not march the next batch on a side stream underneath the current step")
    ap.add_argument("--comm", default="auto", choices=["auto", "f32", "bf16"],
                    help="N>1: dtype the gradient bucket travels in (auto, the default: bf16 when the model reads a bf16 table copy [--table bf16], "
                         "else f32 = the exact mean; bf16 halves xGMI bytes)")
    ap.add_argument("--comm-path", default="rccl", choices=["rccl", "p2p"],
                    help="N>1: how the table's reduce-scatter / all-gather travel: RCCL collectives (default) or round 6's prototype of the "
                         "DIRECT exchange -- one push into the peers' memory (hipIpc) + flags per phase (ngp_hip/p2p.py; needs --comm f32)")
    ap.add_argument("--no-shard", dest="shard", action="store_false",
                    help="N>1: round 1's exchange (ONE all-reduce of the flat gradient bucket + replicated Adam) instead of the default "
                         "reduce-scatter -> Adam on the own 1/N of the table -> all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-configs", dest="configs", action="store_false",
                    help="N = 1, default headline config only: skip the short runs of the other BASELINE configs that are attached "
                         "to the line as `configs` (C2 with a bf16 table, C5 half2, C3 Garden shape, 65 536 rays, initialisation regime)")
    ap.add_argument("--configs-steps", type=int, default=20, help="timed steps of each `configs` entry")
    return ap.parse_args(argv)


class KernelTimer:
    """HIP events around selected launches on the stream they are launched on (torch's current stream);
    nothing is synchronised until after the timed region."""

    def __init__(self):
        self.records = {}
        self.enabled = False
        self.sample = True          # this step's launches get events (bench: every --kernel-events-every-th timed step)

    def wrap(self, ops, name, units_of):
        fn = getattr(ops, name)
        fn = getattr(fn, "_bench_raw", fn)           # measure() may run several times in one process: never nest the wrappers
        timer = self

        def timed(*a, **k):
            if not (timer.enabled and timer.sample):
                return fn(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            timer.records.setdefault(name, []).append((e0, e1, units_of(*a, **k)))
            return out

        timed._bench_raw = fn
        setattr(ops, name, timed)

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [e0.elapsed_time(e1) for e0, e1, _ in recs]
            units = [u for _, _, u in recs]
            out[name] = {"launches": len(ms), "avg_ms": float(np.mean(ms)), "total_ms": float(np.sum(ms)),
                         "avg_units": float(np.mean(units))}
        return out


def cpu_baseline(bits, seconds, table=None, weights=None, targets=None):
    """The oracle (C restatement of the reference's kernels, OpenMP over all host cores) + the same MLPs in fp32
    torch-CPU, timed on 1024-ray batches (BASELINE config 0) of the same workload for ~`seconds`: same ray generator,
    and -- when given -- the bench's own conditioned model state (occupancy bitfield, hash table, MLP weights) and the
    same analytic-scene targets, so the CPU leg sees the sample counts and early termination the GPU leg sees."""
    from oracle import ngp_oracle as ora
    from ngp_hip import synthetic
    ora.build()
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 32))         # tiny GEMMs: more threads only add contention
    n = 1024
    lv = ora.make_levels(2**19, 16, 16, 1024, 2)
    rng = np.random.default_rng(0)
    if table is None:
        table = rng.random(lv.total_entries * 2, dtype=np.float32)
    if weights is None:
        weights = [torch.randn(64, 32) * 0.2, torch.randn(16, 64) * 0.2, torch.randn(64, 32) * 0.2, torch.randn(64, 64) * 0.2,
                   torch.randn(3, 64) * 0.2]
    w = [t.detach().clone().float().requires_grad_(True) for t in weights]
    dtable = np.zeros(lv.total_entries * 2, np.float32)
    done, samples, live = 0, 0, 0
    it = 0
    batches = []
    for b in range(4):                           # rays + targets are made before the clock starts (inputs resident, like the GPU leg)
        o, d = synthetic.lego_rays(n, seed=100 + b)
        tgt = (synthetic.procedural_render_gt(torch.from_numpy(o), torch.from_numpy(d)) if targets == "scene" else torch.rand(n, 3))
        batches.append((o, d, tgt))
    t0 = time.perf_counter()
    while True:
        o, d, target = batches[it % len(batches)]
        noise = rng.random(n, dtype=np.float32)
        hits = ora.ray_aabb(o, d, 0.5)
        rays_a, xyzs, dirs, deltas, ts, S = ora.march_train(o, d, hits, bits, noise, 1, 0.5, 0.0, 128, 1024)
        x01 = (xyzs + 0.5).astype(np.float32)
        enc = torch.from_numpy(ora.hash_fwd_f32(x01, table, lv)).requires_grad_(True)
        dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
        sh = torch.from_numpy(ora.sh16_fwd(((dn + 1) / 2).astype(np.float32)))
        h = torch.relu(enc @ w[0].T) @ w[1].T
        sigma = torch.exp(h[:, 0])
        rgbs = torch.sigmoid(torch.relu(torch.relu(torch.cat([sh, h], 1) @ w[2].T) @ w[3].T) @ w[4].T)
        tot, op, dep, rgb, ws = ora.composite_train_fwd(sigma.detach().numpy(), rgbs.detach().numpy(), deltas, ts, rays_a, 1e-4)
        rgb_t = torch.from_numpy(rgb) + (1 - torch.from_numpy(op))[:, None]
        g_rgb = (2.0 / (3 * n) * (rgb_t - target)).numpy().astype(np.float32)
        g_op = -g_rgb.sum(1)
        ds, dc = ora.composite_train_bwd(g_op, None, g_rgb, None, sigma.detach().numpy(), rgbs.detach().numpy(), deltas, ts,
                                         rays_a, 1e-4)
        torch.autograd.backward([sigma, rgbs], [torch.from_numpy(ds), torch.from_numpy(dc)])
        dtable.fill(0.0)
        ora.hash_bwd_f32_atomic(x01, enc.grad.numpy(), lv, dtable)
        for t in w:
            t.grad = None
        done += n; samples += S; live += int(np.sum(tot)); it += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": done / el, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d batches x 1024 Lego-shape rays (%.1f marched / %.1f composited samples per ray, %s model state), "
                      "oracle/ngp_oracle.c (OpenMP) + fp32 torch-CPU MLP, fwd+bwd without optimizer, %.1f s"
                      % (it, samples / max(done, 1), live / max(done, 1),
                         "the bench's conditioned" if targets == "scene" else "random-init", el)}


class _Probe:
    """What the timing wrappers around the C-ABI entry points (installed once per process) currently record into."""
    timer = None
    event_pool = []
    c_events = {}
    comm = None                 # N > 1: [(e0, e1), ...] around the trainer's collectives on the sampled steps
    only = None                 # entry points that get events (None = all): inside the timed region, the dominant kernel's only


def _install_entry_probes(L):
    """HIP events around the big kernels' launches, on the stream they are launched on.  Installed once; inactive while
    _Probe.event_pool is empty."""
    if getattr(L, "_bench_probed", False):
        return
    L._bench_probed = True

    def wrap_entry(name):
        raw = getattr(L, name)

        def timed(*a):
            t = _Probe.timer
            if t is None or not (t.enabled and t.sample) or not _Probe.event_pool or (_Probe.only is not None and name not in _Probe.only):
                return raw(*a)
            e0, e1 = _Probe.event_pool.pop(), _Probe.event_pool.pop()
            st = torch.cuda.current_stream()
            e0.record(st); rc = raw(*a); e1.record(st)
            _Probe.c_events.setdefault(name, []).append((e0, e1, a))
            return rc
        setattr(L, name, timed)
    for name in TRAINER_KERNELS:
        if hasattr(L, name):
            wrap_entry(name)


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (one per GPU) under
    torch.distributed.run and relay rank 0's JSON line.  Fails loudly when the node has fewer than N devices."""
    import socket
    import subprocess
    one_dev = os.environ.get("NGP_BENCH_ONE_DEVICE", "0") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not one_dev:
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s)" % (args.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # NGP_BENCH_BACKEND=gloo + NGP_BENCH_ONE_DEVICE=1: run the N > 1 code path with every rank on GPU 0 (RCCL refuses two ranks
    # on one device) -- a functional check of the sharding / barrier / max-over-ranks logic on a 1-GPU box, not a measurement
    backend = os.environ.get("NGP_BENCH_BACKEND", "nccl")
    if os.environ.get("NGP_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: no GPU %d on this node (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ngp_hip import lib
    if world > 1:                   # one rank (re)builds the extension if it has to; the others wait instead of racing hipcc
        if rank == 0:
            lib.build()
        dist.barrier()
    lib.build()
    lib.load()
    ctx = {"world": world, "rank": rank, "dev": dev, "backend": backend}
    if os.environ.get("NGP_BENCH_CHILD") == "1" and os.environ.get("NGP_BENCH_CHILD_CRASH") == "1":
        os.abort()                                  # (tests: an isolated leg that dies like a device fault would make it die)
    out = measure(args, ctx)
    if rank == 0 and world == 1 and args.configs and _is_headline(args):
        out["configs"] = other_configs(args, ctx)
    if world > 1 and args.configs and _is_headline(args) and args.comm in ("auto", "f32") and args.shard and "comm_overlap" not in os.environ.get("NGP_EXPERIMENT", ""):
        # ONE invocation decides the multi-GPU defaults: the other exchange variants run in the same process group, behind the headline
        # The variants have never met a real multi-GPU node (no SCALE run in five rounds): the headline must survive one of them
        # hanging.  Every rank arms a watchdog for the variants phase (NGP_BENCH_VARIANT_TIMEOUT seconds, default 240): when it fires,
        # rank 0 prints the line with the records collected so far + `configs_incomplete`, and every rank leaves the process.
        progress = {"configs": [], "current": None, "done": False}
        _variant_watchdog(float(os.environ.get("NGP_BENCH_VARIANT_TIMEOUT", "240")), rank, out, progress)
        cv = comm_variants(args, ctx, out, progress)            # (every rank runs them; rank 0 keeps the records)
        progress["done"] = True
        if rank == 0:
            out["configs"] = cv
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _is_headline(args):
    return (args.scene == "lego" and args.regime == "scene" and args.path == "trainer" and not args.half and args.table == "f32"
            and args.rays in (None, 8192) and not args.graph and args.prefetch)


# the other BASELINE.json configs (and the initialisation regime of SURVEY 8d), as short runs attached to the headline line
OTHER_CONFIGS = [
    ("C2-modules-path", "the headline workload through the reference's own surface: modules.rendering.render + autograd + torch Adam / "
                        "GradScaler / CosineAnnealingLR, the loop train.py:168-201 runs (the drop-in boundary, not FusedTrainer)", ["--path", "modules"]),
    ("C2-bf16-table", "BASELINE config 2 as worded: bf16 storage copy of the fp32 master table", ["--table", "bf16"]),
    ("C5-half2", "BASELINE config 5: half2 encoder (hash_encoder_half semantics) + fp16 MFMA MLP", ["--half"]),
    ("C3-garden", "BASELINE config 3 shape: scale 16, 6 cascades, max_res 4096, 65 536 rays, distortion loss; analytic unbounded scene, model conditioned 512 steps", ["--scene", "garden", "--condition", "512"]),
    ("C2-65536-rays", "the per-GPU batch of BASELINE config 4's global batch on one GPU", ["--rays", "65536", "--pool", "8", "--condition", "512"]),
    ("C2-init-random50", "initialisation regime of SURVEY 8(d): seeded 50 % occupancy, random targets, ~250 samples per ray", ["--regime", "random50"]),
]


def other_configs(args, ctx):
    res = []
    for name, what, extra in OTHER_CONFIGS:
        sub = parse(["--steps", str(args.configs_steps), "--warmup", "5", "--no-cpu-baseline", "--no-configs"] + extra)
        if os.environ.get("NGP_BENCH_CONFIG_CONDITION"):           # (tests: a short conditioning)
            sub.condition = int(os.environ["NGP_BENCH_CONFIG_CONDITION"])
        t0 = time.perf_counter()
        try:
            o = measure(sub, ctx, brief=True)
            roof = o["roofline"] or {}
            res.append({"name": name, "what": what, "args": " ".join(extra), "value": o["value"], "unit": "rays/s", "ms_per_step": o["ms_per_step"],
                        "steps": o["steps"], "warmup": o["warmup"], "dtype": o["dtype"], "rays_per_gpu": o["config"]["rays_per_gpu"],
                        "rm_samples_per_ray": o["rm_samples_per_ray"], "vr_samples_per_ray": o["vr_samples_per_ray"],
                        "live_samples_per_step": o["live_samples_per_step"], "ns_per_live_sample": o.get("ns_per_live_sample"),
                        "shaded_samples_last_step": o.get("shaded_samples_last_step"),
                        "samples_per_sec": o["samples_per_sec"], "path": o["config"]["path"],
                        "grid_updates_in_timed_region": o["config"]["grid_updates_in_timed_region"],
                        "dominant_kernel": roof.get("kernel"), "dominant_kernel_ms": roof.get("avg_launch_ms"), "frac": roof.get("frac"),
                        "bound": roof.get("bound"), "achieved": roof.get("achieved"), "roofline_unit": roof.get("unit"),
                        "kernels_avg_us": {k: round(v["avg_ms"] * 1e3, 1) for k, v in o["kernels"].items()},
                        "wall_seconds_incl_setup": None})
        except Exception as e:                   # a failing side config must not take the headline line with it
            res.append({"name": name, "what": what, "args": " ".join(extra), "error": "%s: %s" % (type(e).__name__, e)})
        res[-1]["wall_seconds_incl_setup"] = time.perf_counter() - t0
    return res


# N > 1: the gradient-exchange variants FusedTrainer offers, each measured in the same process group right after the headline leg
# (which is the first entry: in-line fp32 reduce-scatter / all-gather, sharded optimizer).  A variant = extra arguments + environment.
COMM_VARIANTS = [
    ("bf16-comm+bf16-table", "--comm bf16 --table bf16: the gradient travels as bf16, the parameters come back as the 16-bit copy the "
                             "forward reads (half the bytes both ways)", ["--comm", "bf16", "--table", "bf16"], {}),
    ("no-shard-all-reduce", "--no-shard: SURVEY 8(e)'s single all-reduce of one flat fp32 bucket + replicated Adam (north_star's wording)",
     ["--no-shard"], {}),
    ("overlap-8,0", "NGP_EXPERIMENT comm_overlap=1: the scatter-add issued per level group (8-15, then 0-7), a group's reduce-scatter in flight under "
                    "the next group's launch, all-gathers waited for at the next step's forward", [], {"NGP_EXPERIMENT": "comm_overlap=1;comm_groups=8,0"}),
    # LAST and in CHILD processes (one per rank, their own process group one port up): kernels that write into other devices' memory have
    # never run across xGMI here -- a device fault there must not take the line (headline + the collective variants) with it
    ("p2p-direct", "--comm-path p2p --comm f32: round 6's prototype of SURVEY 8(e)'s direct exchange -- every rank writes slice p of its gradient "
                   "straight into rank p's inbox (peer memory through hipIpc, all links at once, one hop), flags, a local reduce; the updated shards "
                   "travel back the same way.  Runs in child processes of the ranks (isolated: its own conditioning, same state recipe).  UNMEASURED "
                   "on xGMI until this line exists on a multi-GPU node", ["--comm-path", "p2p", "--comm", "f32"], {"NGP_BENCH_ISOLATE": "1"}),
]
COMM_MODEL_BW_GBS = (150.0, 300.0, 450.0)                      # DESIGN.md section 7's bus-bandwidth rows


def _comm_record(name, what, extra, env, o, world):
    """One exchange variant as a `configs` entry: the line's communication fields + the bus bandwidth they imply + DESIGN 7's model."""
    per, nbytes = o.get("comm_breakdown_ms") or {}, o.get("comm_bytes_per_rank_per_step") or {}
    # bus bytes in RCCL's own convention: (N-1)/N of the payload per reduce-scatter / all-gather, twice that per all-reduce
    moved = float(sum(v * (2.0 if k.startswith("all_reduce") else 1.0) for k, v in nbytes.items() if isinstance(v, (int, float))))
    frac = (world - 1) / world
    comm_ms = o.get("comm_ms")
    bus = moved * frac / (comm_ms * 1e-3) / 1e9 if (comm_ms and moved) else None
    stub = o.get("ms_per_step_comm_stubbed")
    model = None
    if stub:
        # T = the measured step without communication + (N-1)/N * bytes / BW for the payload this variant moves
        model = {"%d_GBs" % int(bw): {"ms_per_step": stub + moved * frac / (bw * 1e9) * 1e3,
                                      "rays_per_s": o["config"]["global_batch"] / ((stub + moved * frac / (bw * 1e9) * 1e3) * 1e-3)}
                 for bw in COMM_MODEL_BW_GBS}
    return {"name": name, "what": what, "args": " ".join(extra), "env": env, "value": o["value"], "unit": "rays/s", "n_gpus": world,
            "ms_per_step": o["ms_per_step"], "steps": o["steps"], "comm_ms": comm_ms, "exposed_comm_ms": o.get("exposed_comm_ms"),
            "ms_per_step_comm_stubbed": stub, "comm_breakdown_ms": per, "comm_bytes_per_rank_per_step": nbytes,
            "bus_bandwidth_GBs": bus, "bus_bandwidth_note": "payload bytes x (N-1)/N (x 2 for an all-reduce) / comm_ms (in-line collectives: the "
            "step waits for each; overlapped: what it still waited for, so the figure overstates the link rate there)",
            "design7_model_at_bus_bandwidth": model, "parallelism": o["config"].get("parallelism"),
            "live_samples_per_step": o.get("live_samples_per_step")}


def _variant_watchdog(timeout_s, rank, out, progress):
    import threading

    def run():
        t_end = time.monotonic() + timeout_s
        while time.monotonic() < t_end:
            if progress["done"]:
                return
            time.sleep(0.25)
        if progress["done"]:
            return
        if rank == 0:
            out["configs"] = list(progress["configs"])
            out["configs_incomplete"] = "exchange variant %r did not finish within %.0f s: the line holds the headline leg and the variants completed before it" % (progress["current"], timeout_s)
            print(json.dumps(out), flush=True)
        else:
            time.sleep(3.0)                      # (rank 0 prints first)
        os._exit(0)
    threading.Thread(target=run, daemon=True).start()


def _isolated_leg(args, ctx, extra, timeout_s):
    """One exchange variant in CHILD processes: every rank starts `python bench.py --gpus N <extra>` with its own RANK / LOCAL_RANK and a
    rendezvous one port above the parents'; rank 0 returns its child's line (or raises with the child's exit code and stderr tail)."""
    import subprocess
    world, rank = ctx["world"], ctx["rank"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC") and k != "NGP_BENCH_ISOLATE"}
    # the children's rendezvous: a port rank 0 finds free now, told to every rank through the parents' group (rank 0's child hosts the store)
    port = [0]
    if rank == 0:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0, device=ctx["dev"] if ctx.get("backend") == "nccl" else None)
    env["MASTER_PORT"] = str(port[0])
    env["NGP_BENCH_CHILD"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--condition", str(args.condition), "--kernel-events-every", str(args.kernel_events_every), "--no-cpu-baseline", "--no-configs"] + extra
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        raise RuntimeError("isolated leg: the child of rank %d did not finish within %.0f s" % (rank, timeout_s))
    if rank != 0:
        return None
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    if res.returncode != 0 or not lines:
        raise RuntimeError("isolated leg: child exit code %s; stderr tail: %s" % (res.returncode, res.stderr[-600:]))
    return json.loads(lines[-1])


def comm_variants(args, ctx, headline, progress=None):
    world, rank = ctx["world"], ctx["rank"]
    res = progress["configs"] if progress is not None else []
    if rank == 0:
        res.append(_comm_record("inline-f32 (headline)", "the line above: reduce-scatter(AVG) fp32 -> Adam on the own 1/N -> all-gather fp32, "
                                "in line on the step's stream", [], {}, headline, world))
    for name, what, extra, env in COMM_VARIANTS:
        sub = parse(["--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup), "--condition", str(args.condition),
                     "--kernel-events-every", str(args.kernel_events_every), "--no-cpu-baseline", "--no-configs"] + extra)
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        if progress is not None:
            progress["current"] = name
        try:
            if env.get("NGP_BENCH_ISOLATE") == "1":
                err = None
                try:
                    o = _isolated_leg(sub, ctx, extra, float(os.environ.get("NGP_BENCH_ISOLATED_TIMEOUT", "150")))
                except RuntimeError as e:              # (per rank: the parents stay in step through the barrier below)
                    o, err = None, e
                if rank == 0 and err is not None:
                    raise err
            else:
                o = measure(sub, ctx)
            if rank == 0:
                res.append(_comm_record(name, what, extra, env, o, world))
        except Exception as e:                       # (raised on every rank alike: argument / setup errors)
            if rank == 0:
                res.append({"name": name, "what": what, "args": " ".join(extra), "env": env, "error": "%s: %s" % (type(e).__name__, e)})
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        dist.barrier()
    return res


def comm_fields(trainer, ctx, args):
    """N > 1: what the gradient exchange cost on the sampled steps (HIP events on the step's stream around every collective of
    FusedTrainer: they are issued in line, so the step waits for each of them -- only the next batch's march, on its side
    stream, runs underneath), and which RCCL this was."""
    recs = _Probe.comm or []
    per = {}
    for name, e0, e1 in recs:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    n_steps = max((len(v) for v in per.values()), default=0)
    comm_ms = sum(float(np.sum(v)) for v in per.values()) / n_steps if n_steps else None
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    from ngp_hip import experiment as _exp
    overlap = trainer is not None and getattr(trainer, "_groups", None) is not None
    stub_ms = getattr(_Probe, "stub_ms", None)
    return {"comm_ms": comm_ms,
            "exposed_comm_ms": None if (stub_ms is None or ctx.get("ms_per_step") is None) else ctx["ms_per_step"] - stub_ms,
            "ms_per_step_comm_stubbed": stub_ms,
            "comm_overlap": {"enabled": overlap, "level_groups": _exp.get("comm_groups", "8,0") if overlap else None},
            "comm_breakdown_ms": {k: float(np.mean(v)) for k, v in per.items()},
            "comm_note": "comm_ms = per step, HIP events on the step's stream around each collective (sampled steps) -- in line: the "
                         "transfer; with NGP_EXPERIMENT comm_overlap=1: what the step's stream WAITS for the collective issued earlier.  "
                         "exposed_comm_ms = ms_per_step minus the same K steps re-run with every collective replaced by its local "
                         "part (ms_per_step_comm_stubbed): measured, not assumed",
            "comm_bytes_per_rank_per_step": None if trainer is None else trainer.comm_bytes_per_step(),
            "rccl_version": ver, "rccl_ranks": ctx["world"], "backend": ctx["backend"],
            "rccl_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}}


def measure(args, ctx, brief=False):
    """One configuration: build the model + trainer, condition, warm up, time args.steps steps.  Returns the line's dictionary on
    rank 0 (None elsewhere)."""
    try:
        return _measure(args, ctx, brief)
    finally:
        # several configurations run in one process (`configs`): give the arena-sized buffers of this one back before the next
        _Probe.timer, _Probe.event_pool, _Probe.c_events, _Probe.comm, _Probe.only = None, [], {}, None, None
        from ngp_hip.fused import TrainArena
        TrainArena._cache.clear()
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()


def _measure(args, ctx, brief):
    world, rank, dev = ctx["world"], ctx["rank"], ctx["dev"]
    from ngp_hip import lib, ops, synthetic
    from ngp_hip.dist import GradReducer
    from modules.networks import NGP
    from modules.rendering import MAX_SAMPLES, render

    timer = KernelTimer()
    timer.wrap(ops, "hash_fwd_f32", lambda xyzs, table, lv: xyzs.shape[0])
    timer.wrap(ops, "hash_bwd_f32", lambda xyzs, dout, lv, dtable: xyzs.shape[0])
    use_trainer = args.path == "trainer"

    torch.manual_seed(23)                       # identical replicas on every rank (train.py:39-42 uses 23)
    np.random.seed(23)
    if args.half and args.table != "f32":
        raise SystemExit("--half already selects the fp16 table")
    garden = args.scene == "garden"
    scene = args.regime == "scene"               # (round 4: Garden has an analytic scene too, synthetic.garden_field: C3 is conditioned like C2)
    if scene and args.condition % 16 != 0:
        raise SystemExit("--condition must be a multiple of 16 (the occupancy-update cadence)")
    if args.rays is None:
        args.rays = 65536 if garden else 8192
    esf = 1.0 / 256 if garden else 0.0                                       # train.py:54
    w_dist = 1e-3 if garden else 0.0                                         # opt.py:77-83 "1e-3 for real scene"
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):         # the encoder announces itself like the reference's does: keep stdout = the JSON line
        model = NGP(scale=16.0 if garden else 0.5, max_res=4096 if garden else 1024, half_opt=args.half,
                    table_dtype=torch.bfloat16 if args.table == "bf16" else None).to(dev)
    golden = os.path.join(ROOT, "tests", "golden", "lego_density_bitfield.npz")
    bits = bits_np = None
    if garden and not scene:
        bits_np = synthetic.ball_slab_bitfield(model.cascades, 16.0, seed=23)
    elif scene:
        pass                                     # the model's own occupancy grid
    elif args.regime == "lego":
        bits_np = np.load(golden)["density_bitfield"]
    elif args.regime == "random50":
        bits_np = synthetic.random_bitfield(1, fraction=0.5, seed=23)
    elif args.regime == "ones":
        bits_np = np.full(128**3 // 8, 255, np.uint8)
    if bits_np is not None:
        bits = torch.from_numpy(bits_np).to(dev)
        model.density_bitfield.copy_(bits)

    trainer = None
    opt_name = None
    if use_trainer:
        from ngp_hip.trainer import FusedTrainer
        trainer = FusedTrainer(model, lr=1e-2, max_steps=20000, init_scale=2.0**16 if args.half else 2.0**19, world_size=world,
                               exp_step_factor=esf, distortion_loss_w=w_dist,
                               grad_comm_dtype={"bf16": torch.bfloat16, "f32": torch.float32, "auto": None}[args.comm],
                               shard_optimizer=args.shard if world > 1 else None, exchange=args.comm_path)
    else:
        # train.py:143-156 picks apex.optimizers.FusedAdam when `import apex` works and torch.optim.Adam otherwise.  With this package's
        # compat/ directory on the path (how scripts/run_reference_train.py runs the unchanged driver) the import resolves to
        # compat/apex: one multi-tensor launch on ngp_adam_multi that takes GradScaler's scale and inf flag on the device.
        # NGP_BENCH_TORCH_ADAM=1: the driver's fall-back, torch.optim.Adam(fused=True).
        opt_name = "torch.optim.Adam(fused=True)"
        opt = None
        if os.environ.get("NGP_BENCH_TORCH_ADAM", "0") != "1":
            compat_dir = os.path.join(ROOT, "taichi-nerfs_amd", "compat")
            if compat_dir not in sys.path:
                sys.path.append(compat_dir)
            try:
                import apex
                opt = apex.optimizers.FusedAdam(model.parameters(), lr=1e-2, eps=1e-15)
                opt_name = "apex.optimizers.FusedAdam (compat/apex: ngp_adam_multi)"
            except ImportError:
                opt = None
        if opt is None:
            opt = torch.optim.Adam(model.parameters(), 1e-2, eps=1e-15, fused=True)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 20000, 1e-2 / 30)
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0**16 if args.half else 2.0**19)
        reducer = GradReducer(model, world) if world > 1 else None

    # a pool of synthetic batches resident in HBM before the timed region (rank-dependent shards of one stream of seeds);
    # scene regime: every ray's target colour is the analytic scene's radiance along it, rendered here once
    n_pool = (min(args.pool, 8) if garden else args.pool) if scene else 8       # (Garden: 65 536 rays per batch)
    pool = []
    for b in range(n_pool):
        o, d = (synthetic.garden_rays if garden else synthetic.lego_rays)(args.rays, seed=1000 + 97 * b + rank)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        if scene and garden:
            tgt = synthetic.garden_render_gt(o, d, scale=16.0).contiguous()
        elif scene:
            tgt = synthetic.procedural_render_gt(o, d).contiguous()
        else:
            g = torch.Generator(device="cpu").manual_seed(b * 131 + rank)
            tgt = torch.rand(args.rays, 3, generator=g).to(dev)
        pool.append((o, d, tgt))

    n_extra = args.steps + 8                                  # the no-prefetch leg after the timed region logs too
    state = {"rm": 0, "vr": 0, "k": 0}
    # sample counts are informational: they are copied out on every STAT_EVERY-th step only (a 4-byte device-to-device copy
    # is a ~4.5 us kernel on the step's stream); 7 is coprime with the pool size, so every batch is sampled
    STAT_EVERY = 7
    n_log = (args.steps + args.warmup + n_extra + 4) // STAT_EVERY + 1
    stat_log = torch.zeros(n_log, 1, device=dev, dtype=torch.int32)
    vr_log = torch.zeros(n_log, 1, device=dev, dtype=torch.int32)
    live_log = torch.zeros(n_log, 1, device=dev, dtype=torch.int32)

    # the trainer launches go straight through the C ABI: HIP events around the big kernels, on the launch stream
    c_events = {}
    # events are created up front (hipEventCreate inside the timed loop costs more than the kernels it would time)
    event_pool = ([torch.cuda.Event(enable_timing=True) for _ in range(2 * 10 * (args.steps + args.warmup + 2))]
                  if use_trainer and not args.graph and args.kernel_events else [])
    _Probe.timer, _Probe.event_pool, _Probe.c_events, _Probe.comm = timer, event_pool, c_events, []
    if use_trainer and not args.graph:
        _install_entry_probes(lib.load())
    if use_trainer and world > 1:
        trainer.comm_probe = lambda: (timer.enabled and timer.sample, _Probe.comm)

    thr = 0.01 * MAX_SAMPLES / 3**0.5                        # train.py:180

    def grid_update(i):
        """train.py:178-182: every 16 steps, all cells during the first 256 steps.  Scene regime: the model's own grid is what
        the next steps march.  Diagnostic regimes: the update's full cost is paid, then the fixed bitfield is put back."""
        if use_trainer:
            trainer.update_density_grid(thr, warmup=scene and i < 256)
        else:
            model.update_density_grid(thr, warmup=scene and i < 256)
        if not scene:
            model.density_bitfield.copy_(bits)

    ev_every = max(1, min(args.kernel_events_every, args.steps // 3))        # (a short run still samples three steps)

    def trainer_step(i, prefetch=True, log=True):
        rays_o, rays_d, target = pool[i % n_pool]
        # per-kernel events on every ev_every-th step; an occupancy-update step (its 1 M-point encodes go through the same entry
        # points) hands its turn to the next step, so that the per-kernel averages are those of the training step's launches
        ee = state.get("ev_every", ev_every)
        due = state["k"] % ee == ee // 2 or state.get("ev_owed", False)
        timer.sample = due and i % 16 != 0
        state["ev_owed"] = due and i % 16 == 0
        if i % 16 == 0:
            grid_update(i)
        nxt = pool[(i + 1) % n_pool]
        pre = (nxt[0], nxt[1]) if (prefetch and (i + 1) % 16 != 0) else None      # never across a grid update
        out = trainer.step(rays_o, rays_d, target, prefetch=pre)
        k = state["k"]
        if log and k % STAT_EVERY == 0 and k // STAT_EVERY < stat_log.shape[0]:
            stat_log[k // STAT_EVERY, 0].copy_(out["rm_samples"][0])
            if trainer.live_backward:
                # the live samples ARE the composited ones (the first vr[r] samples of every ray): their count is on the device already --
                # no reduction kernel + copy on the step's stream for a number the line only reports
                vr_log[k // STAT_EVERY, 0].copy_(trainer._live_total[0])
            else:
                vr_log[k // STAT_EVERY, 0] = out["vr_per_ray"].sum()
            live_log[k // STAT_EVERY, 0].copy_(trainer._live_total[0])
        state["k"] = k + 1

    def step(i, prefetch=True, log=True):
        if use_trainer:
            return trainer_step(i, prefetch, log)
        rays_o, rays_d, target = pool[i % n_pool]
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            if i % 16 == 0:
                grid_update(i)
            res = render(model, rays_o, rays_d, exp_step_factor=esf)
            loss = F.mse_loss(res["rgb"], target)
            if w_dist > 0:
                from modules.distortion import distortion_loss
                loss = loss + w_dist * distortion_loss(res).mean()
        opt.zero_grad()
        scaler.scale(loss).backward()
        if reducer is not None:
            reducer.all_reduce()
        scaler.step(opt)
        scaler.update()
        sched.step()
        # train.py:203-221 reads rm_samples / vr_samples in its log line only (every 1000th step); reading them on every step would
        # put two reductions + two adds on the step's stream that the reference's loop does not have.  Sampled like the trainer path.
        k = state["k"]
        if log and k % STAT_EVERY == 0 and k // STAT_EVERY < stat_log.shape[0]:
            stat_log[k // STAT_EVERY, 0] = res["rm_samples"]
            vr_log[k // STAT_EVERY, 0] = res["vr_samples"]
        state["k"] = k + 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed conditioning: the model becomes a (partly) trained model of the scene its targets come from --------------
    base = 0
    t_cond = 0.0
    det_cond = bool(scene and use_trainer and args.det_condition and args.condition > 0)
    if scene:
        fence()
        t0 = time.perf_counter()
        if det_cond:
            trainer.set_deterministic(True)
        for i in range(args.condition):
            step(i, prefetch=args.prefetch, log=False)
        if det_cond:
            trainer.set_deterministic(False)
        fence()
        t_cond = time.perf_counter() - t0
        base = args.condition
    if not os.environ.get("NGP_BENCH_KEEP_GC"):
        # a generation-2 pass of the cyclic collector over the process's long-lived objects takes 36 ms here (measured with
        # --step-trace), 55 steps' worth; where the next one lands depends on the allocation count so far.  Collect now and freeze
        # the survivors so a 20-step timed region cannot contain one (young passes: 0.02-0.6 ms) -- BEFORE the warm-up steps, so
        # that the GPU does not sit idle for those 36 ms right in front of the timed region
        gc.collect()
        gc.freeze()
    state["k"] = 0
    # Per-kernel HIP events cost a barrier packet each (~6 us of queue time): with all seven kernels bracketed, a sampled step is
    # ~80 us longer.  So: the WARM-UP steps (untimed, same state) carry events on every kernel -- the `kernels` / `rooflines`
    # tables come from them -- and the kernel that dominates there is the only one bracketed inside the timed region, whose
    # events give `roofline` (the contract's live measurement over the timed region).
    timer.enabled = True
    state["ev_every"] = 1
    for i in range(base, base + args.warmup):
        # warm-up steps are the timed steps, sample-count logging included: the first torch reduction of a process loads its
        # code object (13-55 ms, host blocked), and since the grid update stopped using torch's scans that first call was the
        # log of the first timed step
        step(i, prefetch=args.prefetch, log=True)
    if use_trainer and args.graph and world == 1:
        trainer.capture(args.rays)
    state["rm"] = 0; state["vr"] = 0; state["k"] = 0; state["ev_every"] = ev_every; state["ev_owed"] = False
    fence()
    warm_events = {k_: list(v_) for k_, v_ in c_events.items()}
    warm_live = int(trainer._live_total[0]) if (use_trainer and trainer.live_backward) else None
    c_events.clear()
    if use_trainer and warm_events:
        tot_ms = {}
        for name_, evs_ in warm_events.items():
            key_ = TRAINER_KERNELS[name_][0]
            if key_ == "march_count" and args.prefetch and not args.graph:
                continue                                      # runs on the side stream underneath the step: not on the critical path
            tot_ms[key_] = tot_ms.get(key_, 0.0) + sum(e0.elapsed_time(e1) for e0, e1, _ in evs_)
        dom_key = max(tot_ms, key=tot_ms.get)
        _Probe.only = {n_ for n_ in TRAINER_KERNELS if TRAINER_KERNELS[n_][0] == dom_key}
    timer.enabled = True
    trace = [] if args.step_trace else None
    if trace is not None:
        import cProfile, pstats, io                            # (first timed step is profiled on the host: stalls show up there)
    gc_log = []
    if trace is not None:
        def _gc_cb(phase, info, _s=[0.0]):
            if phase == "start":
                _s[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], (time.perf_counter() - _s[0]) * 1e3, len(trace)))
        gc.callbacks.append(_gc_cb)
    t0 = time.perf_counter()
    for i in range(base + args.warmup, base + args.warmup + args.steps):
        if trace is not None:                                   # (diagnostic: one event + one host stamp per step)
            ev = torch.cuda.Event(enable_timing=True); ev.record(); trace.append((i, time.perf_counter(), ev))
        if trace is not None and len(trace) == 1 and rank == 0:
            pr = cProfile.Profile(); pr.enable()
            step(i, prefetch=args.prefetch)
            pr.disable()
            buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(12)
            open(args.step_trace + ".first_step_profile.txt", "w").write(buf.getvalue())
            continue
        step(i, prefetch=args.prefetch)
    if trace is not None:
        ev = torch.cuda.Event(enable_timing=True); ev.record(); trace.append((-1, time.perf_counter(), ev))
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if trace is not None and rank == 0:
        with open(args.step_trace, "w") as f:
            f.write("# step  step%16  host_ms(launch loop)  device_ms(event to event)\n")
            for (i, th, ev), (_, th2, ev2) in zip(trace[:-1], trace[1:]):
                f.write("%d %d %.4f %.4f\n" % (i, i % 16, (th2 - th) * 1e3, ev.elapsed_time(ev2)))
            for gen, ms, at in gc_log:
                f.write("# gc generation %d: %.3f ms during timed step %d\n" % (gen, ms, at - 1))
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    k_timed = state["k"]

    # ---- after the timed region (untimed, informational): the same K steps again with the march NOT prefetched, i.e. what a
    # loop pays whose next batch is not known one step ahead
    elapsed_np = None
    if use_trainer and args.prefetch and not args.graph and world == 1 and not brief:
        i0 = base + args.warmup + args.steps
        i0 += (-i0) % 16                                                    # same phase relative to the grid updates ...
        i0 += (base + args.warmup) % 16                                     # ... as the timed region
        step(i0 - 1, prefetch=False, log=False)
        fence()
        t0 = time.perf_counter()
        for i in range(i0, i0 + args.steps):
            step(i, prefetch=False, log=False)
        fence()
        elapsed_np = time.perf_counter() - t0

    # ---- N > 1, after the timed region (untimed): the same K steps with every collective replaced by its local part (FusedTrainer
    # `_comm_stub`): ms_per_step minus this is the communication the step really waits for (round 3 reported comm_ms as exposed)
    elapsed_stub = None
    # (the stubbed steps below let the ranks drift apart: everything the line reports about the model's state is read before them)
    loss_at_end = trainer.last_loss() if (use_trainer and scene) else None
    if use_trainer and world > 1 and not args.graph and not brief:
        i0 = base + args.warmup + args.steps
        i0 += (-i0) % 16
        i0 += (base + args.warmup) % 16
        trainer.finish_comm()
        trainer._comm_stub = True
        step(i0 - 1, prefetch=args.prefetch, log=False)
        fence()
        t0 = time.perf_counter()
        for i in range(i0, i0 + args.steps):
            step(i, prefetch=args.prefetch, log=False)
        fence()
        elapsed_stub = time.perf_counter() - t0
        trainer._comm_stub = False
        tt = torch.tensor([elapsed_stub], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_stub = float(tt.item())
    _Probe.stub_ms = None if elapsed_stub is None else elapsed_stub / args.steps * 1e3

    # both paths sample the counts on every STAT_EVERY-th step: scaled back to the K timed steps
    n_st = (k_timed + STAT_EVERY - 1) // STAT_EVERY
    state["rm"] = stat_log[:n_st].sum(dtype=torch.int64) * k_timed // max(n_st, 1)
    state["vr"] = vr_log[:n_st].sum(dtype=torch.int64) * k_timed // max(n_st, 1)
    rm = int(state["rm"]); vr = int(state["vr"])
    total_rays = args.rays * world * args.steps
    out = None
    grid_updates = sum(1 for i in range(base + args.warmup, base + args.warmup + args.steps) if i % 16 == 0)
    if rank == 0:
        ks = timer.summary()
        rooflines = {}
        live_avg = None
        gaps = {}
        if use_trainer:
            marched = rm / max(args.steps, 1)                  # marched samples per step (launches are sized for the arena)
            live_avg = (float(live_log[:n_st].sum(dtype=torch.int64)) / max(n_st, 1)) if trainer.live_backward else marched
            shaded_last = trainer.shaded_samples()             # chunked forward: what the last timed step shaded (None: everything)
            # the dense Adam pass skips float4 groups that never received a gradient (g = m = v = 0: exact fixed points) after
            # reading g, m, v; everything else reads p too and writes p, m, v and the zeroed g.  Bytes it really moves:
            # (sharded optimizer: this rank's pass covers its 1/N of the table only)
            lo_ = trainer.rank * trainer.shard_len if trainer.shard else 0
            hi_ = lo_ + trainer.shard_len if trainer.shard else trainer.nt_pad
            n4 = (hi_ - lo_) // 4
            touched4 = int(((trainer.table_m[lo_:hi_].view(-1, 4) != 0) | (trainer.table_v[lo_:hi_].view(-1, 4) != 0)).any(1).sum())
            adam_bytes = 48.0 * n4 + 80.0 * touched4 + (8.0 * touched4 if (args.half or args.table == "bf16") else 0.0)
            copy16_b = 2.0 if (args.half or args.table == "bf16") else 0.0

            def adam_range_bytes(n_):                          # the same count for a launch over the first n_ floats of the table
                n_ = int(n_)
                n4_ = n_ // 4
                t4_ = int(((trainer.table_m[:n_].view(-1, 4) != 0) | (trainer.table_v[:n_].view(-1, 4) != 0)).any(1).sum()) if n4_ else 0
                return 48.0 * n4_ + 80.0 * t4_ + 4.0 * copy16_b * t4_
            # round 5: the part of the table whose optimizer runs in the scatter-add's flush -- m, v, p read for every parameter, p, m, v
            # [+ the 16-bit copy] written for the touched ones
            npre_ = trainer._adam_prefix.get(0) if getattr(trainer, "_flush_adam", False) else None
            flush_bytes = 0.0
            if npre_ is not None and npre_ >= 0:
                n_fl = trainer.nt - npre_
                t_fl = int(((trainer.table_m[npre_:trainer.nt] != 0) | (trainer.table_v[npre_:trainer.nt] != 0)).sum())
                flush_bytes = 12.0 * n_fl + (12.0 + copy16_b) * t_fl
            # critical-path gaps on the sampled steps (each includes the two event packets in between): prepass end -> MLP backward
            # start, MLP backward end -> scatter-add start
            ev_b = warm_events.get("ngp_mlp_bwd_live_parts", []) or warm_events.get("ngp_mlp_bwd_live", []); ev_m = warm_events.get("ngp_hash_bwd_sliced_main_adam_step", []) or warm_events.get("ngp_hash_bwd_sliced_main_adam", []) or warm_events.get("ngp_hash_bwd_sliced_main_slabs", []) or warm_events.get("ngp_hash_bwd_sliced_main", [])
            ev_p = warm_events.get("ngp_hash_bwd_sliced_prep", [])
            if ev_b and len(ev_b) == len(ev_m) == len(ev_p):
                gaps["mlp_bwd_end_to_scatter_start_us"] = float(np.mean([b[1].elapsed_time(m[0]) for b, m in zip(ev_b, ev_m)])) * 1e3
                gaps["prep_end_to_mlp_bwd_start_us"] = float(np.mean([p_[1].elapsed_time(b[0]) for p_, b in zip(ev_p, ev_b)])) * 1e3
            agg = {}                                           # key -> [launches, total_ms, total_work, bound, per_unit, unit, units]
            timed_names = {n_ for n_, e_ in c_events.items() if e_}
            all_events = dict(warm_events)
            all_events.update({n_: c_events[n_] for n_ in timed_names})        # the dominant kernel: its timed-region events
            measured_in = {}
            for name, evs in all_events.items():
                measured_in[TRAINER_KERNELS[name][0]] = ("timed region" if name in timed_names else
                                                         "warm-up steps (untimed, same state, every kernel bracketed)")
                key, bound, per_unit, unit = TRAINER_KERNELS[name]
                for e0, e1, a in evs:
                    if name == "ngp_adam_step" and a[4] < 1000000:
                        continue                               # the 9 408-weight pass: keep the table pass only
                    if unit == "param":
                        units = float(a[5] if name == "ngp_adam_all_ex" else a[4])
                    elif unit == "ray":
                        units = float(args.rays)
                    elif unit == "n_arg":
                        units = float(a[3])
                    elif unit == "live":
                        units = float(live_avg)
                    elif unit == "shaded":
                        units = float(shaded_last or 0) / max(len(trainer._chunk_rounds), 1)
                    else:
                        # _ex launches: device-side count (the marched samples of the step) unless n_dev is NULL
                        # (occupancy-update encodes: exact n = arg 3)
                        n_dev = a[4] if name in ("ngp_hash_fwd_f32_ex", "ngp_hash_fwd_bf16_ex", "ngp_hash_fwd_f16_ex", "ngp_mlp_fwd_ex") else None
                        units = float(a[3]) if (n_dev is not None and getattr(n_dev, "value", 1) is None) else float(marched)
                    if unit == "param":
                        work = adam_bytes if units >= n4 * 4 else adam_range_bytes(units)
                    else:        # (the scatter-add with the optimizer in its flush too: the headline prices it with 8(d)'s 2188 B per
                        # live sample ALONE, so that work_per_unit x avg_units_per_launch / avg_launch_ms reproduces `achieved` and
                        # rounds stay comparable; the optimizer's bytes are in the `optimizer_in_flush` sub-record)
                        work = per_unit * units + ((40 if name == "ngp_march_train_fused" else 8) * marched if key == "march_count" else 0)
                    rec = agg.setdefault(key, [0, 0.0, 0.0, bound, per_unit, "sample" if unit in ("n_arg", "live", "shaded") else unit, 0.0])
                    rec[0] += 1; rec[1] += e0.elapsed_time(e1); rec[2] += work; rec[6] += units
            for key, (n_l, tot_ms, tot_work, bound, per_unit, unit, tot_units) in agg.items():
                ks[key] = {"launches": n_l, "avg_ms": tot_ms / n_l, "total_ms": tot_ms, "avg_units": tot_units / n_l}
                ach = tot_work / (tot_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
                peak = HBM_PEAK_GBS if bound == "hbm" else MFMA_PEAK_TFLOPS
                rooflines[key] = {"kernel": key, "bound": bound, "achieved": float(ach), "peak": peak,
                                  "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": float(ach / peak),
                                  "traffic": None, "work_per_unit": per_unit, "unit_of_work": unit,
                                  "avg_units_per_launch": tot_units / n_l, "avg_launch_ms": tot_ms / n_l, "launches": n_l,
                                  "measured_in": measured_in.get(key)}
                if key.startswith("adam"):
                    rooflines[key]["note"] = ("bytes = what the pass really moves: 48 B per float4 group (g, m, v read) + 80 B per TOUCHED "
                                              "group (p read; p, m, v, zeroed g written); %d of %d groups touched" % (touched4, n4))
                    rooflines[key]["work_per_unit"] = adam_bytes / max(tot_units / n_l, 1)
        else:
            for key, k in ks.items():
                bps = BYTES_PER_SAMPLE[key]
                ach = bps * k["avg_units"] / (k["avg_ms"] * 1e-3) / 1e9
                rooflines[key] = {"kernel": key, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBS, "traffic": None, "work_per_unit": bps, "unit_of_work": "sample",
                                  "avg_units_per_launch": k["avg_units"], "avg_launch_ms": k["avg_ms"], "launches": k["launches"]}
        # PMC traffic (HBM bytes per launch) comes from separate rocprofv3 --pmc passes of THIS command (profiles/r02_pmc.json,
        # which records the workload state it was taken in); it is attached only if that state matches this run within 15 %
        # (two runs of the same command end their conditioning 5-12 % apart in live samples per step: float-atomic order in the
        # MLP weight gradients), otherwise traffic stays null -- a counter value from another state says nothing about this one
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json")) if os.path.exists(q)),
                        os.path.join(ROOT, "profiles", "r05_pmc.json"))
        pmc_name = "profiles/" + os.path.basename(pmc_path)
        traffic_src = None
        if "march_count" in rooflines and use_trainer and args.prefetch and not args.graph:
            rooflines["march_count"]["note"] = (
                "launched on the side stream under the scatter-add's tail and the optimizer pass: the event-to-event time includes waiting "
                "for the scatter-add's workgroups to retire (their 128-VGPR waves fill the register files) and the slowdown of running "
                "beside the optimizer; alone the launch takes ~45-55 us (rocprofv3 / profiles/microbench/march_waves.py)")
        if "hash_bwd_f32" in rooflines and use_trainer and trainer.hash_bwd == "sliced" and flush_bytes > 0:
            r_ = rooflines["hash_bwd_f32"]
            with_opt = float((r_["work_per_unit"] * r_["avg_units_per_launch"] + flush_bytes) / (r_["avg_launch_ms"] * 1e-3) / 1e9)
            r_["optimizer_in_flush"] = {
                "bytes_per_launch": flush_bytes, "parameters": trainer.nt - npre_,
                "with_optimizer_bytes": {"achieved": with_opt, "frac": with_opt / HBM_PEAK_GBS},
                "note": "this launch is the scatter-add AND torch.optim.Adam for the table levels whose slices have one owner (92 % of the "
                        "C2 table).  The record's `achieved` / `frac` price the launch with the scatter-add's 2188 B per live sample alone "
                        "(constant accounting across rounds; round 5 printed the with-optimizer figure as the headline); "
                        "`with_optimizer_bytes` adds the optimizer bytes the flush really moves (12 B per parameter read, 12 B per touched "
                        "parameter written) over the same launch time"}
        if "hash_bwd_f32" in rooflines and use_trainer and trainer.hash_bwd == "sliced":
            rooflines["hash_bwd_f32"]["note"] = (
                "bytes = SURVEY 8(d)'s algorithmic figure for the reference's autodiff scatter (2188 B per live sample: position, "
                "gradient row, 16 x 8 corner read-modify-writes); the LDS-sliced kernel keeps the read-modify-write in LDS, so its HBM "
                "traffic (PMC, `traffic`) is below that figure: `frac` is the contract's algorithmic-bytes fraction, not HBM utilisation")
        if os.path.exists(pmc_path) and use_trainer:
            pmc = json.load(open(pmc_path))
            st = pmc.get("state", {})
            same = (st.get("regime") == args.regime and st.get("rays") == args.rays and not args.half and args.table == "f32"
                    and abs(st.get("live_samples_per_step", -1) - live_avg) <= 0.15 * max(live_avg, 1)
                    and abs(st.get("marched_samples_per_step", -1) - marched) <= 0.15 * max(marched, 1))
            if same:
                traffic_src = ("%s (separate --pmc passes of this command at %.0f live / %.0f marched samples per step; "
                               "this run: %.0f / %.0f)" % (pmc_name, st["live_samples_per_step"], st["marched_samples_per_step"], live_avg, marched))
                for key, r in rooflines.items():
                    if key in pmc.get("kernels", {}):
                        r["traffic"] = pmc["kernels"][key]["hbm_bytes_per_launch"]
                        r["traffic_source"] = traffic_src
        # dominant kernel = largest total time on the step's critical path; with --prefetch the march of the next batch
        # runs on a side stream underneath the other kernels (15 of 16 steps), so it is reported but not eligible
        eligible = [k for k in rooflines if not (k == "march_count" and use_trainer and args.prefetch and not args.graph)
                    ]
        dom = max(eligible, key=lambda k: ks[k]["avg_ms"], default=None)     # (one launch of each per step: the per-launch average ranks them)
        roof = rooflines.get(dom)
        # the reference's ONE backward op (hash_encoder.py:269) is TWO launches here -- the prepass (hit bitmaps +
        # compact positions, in line before the MLP backward) and the main launch.  The same algorithmic bytes over the SUM of the
        # two durations (the prepass from the warm-up steps' events, the main launch from the timed region)
        if roof is not None and dom == "hash_bwd_f32" and "hash_bwd_prep" in ks and use_trainer and trainer.hash_bwd == "sliced":
            both_ms = roof["avg_launch_ms"] + ks["hash_bwd_prep"]["avg_ms"]
            ach2 = roof["work_per_unit"] * roof["avg_units_per_launch"] / (both_ms * 1e-3) / 1e9
            roof["with_prepass"] = {"avg_ms_main_plus_prep": both_ms, "achieved": float(ach2), "frac": float(ach2 / HBM_PEAK_GBS),
                                    "note": "the reference's one backward op = prepass + main launch here; same algorithmic bytes over both"}
        workload = {"regime": args.regime, "rm_samples_per_ray": rm / total_rays * world, "vr_samples_per_ray": vr / total_rays * world,
                    "live_over_marched": vr / max(rm, 1),
                    "occupied_fraction": float((torch.cat([(model.density_bitfield >> b) & 1 for b in range(8)]) > 0).float().mean())}
        if scene:
            workload.update({"targets": ("analytic Garden-shape unbounded scene (ngp_hip/synthetic.py: object on a table, ground to 0.85 x scale, "
                                         "boxes at radii 1.5-10), exponentially spaced integration per ray, black background") if garden else
                                        "analytic Lego-shape scene (ngp_hip/synthetic.py), dense-integration radiance per ray, white background",
                             "conditioning_steps": args.condition, "conditioning_seconds": t_cond, "pool_batches": n_pool,
                             "conditioning_mode": ("deterministic (ray-order packing, one owner per table slice, no float atomics; the warm-up "
                                                   "and timed steps run the default path)") if det_cond else "default path",
                             "occupancy": "the model's own grid (update every 16 steps; all-cell warm-up for steps < 256)"})
            if use_trainer:
                workload["loss_at_end"] = loss_at_end
        if garden:
            text = ("360_v2 Garden shape (BASELINE C3): %d rays/GPU/step, scale 16, 6 cascades 128^3, hash grid L=16 F=2 "
                    "T=2^19 max_res=4096 (%s table), exp_step_factor 1/256, %s, "
                    "MSE + 1e-3 distortion loss, full train step (fwd+bwd+GradScaler+Adam, grid update every 16 steps); "
                    "%.1f marched / %.1f composited samples per ray"
                    % (args.rays, "f16" if args.half else args.table,
                       ("analytic-scene targets, model conditioned for %d steps, marching its own occupancy grid" % args.condition) if scene
                       else "synthetic ball+slab+far-cells occupancy, random target colours (diagnostic state)",
                       rm / total_rays * world, vr / total_rays * world))
        else:
            text = ("Synthetic-NeRF Lego shape (BASELINE C2%s): %d rays/GPU/step, scale 0.5, 1 cascade 128^3, hash grid L=16 F=2 T=2^19 "
                    "max_res=1024 (%s table), %s, full train step (fwd+bwd+GradScaler+Adam, grid update every 16 steps); "
                    "%.1f marched / %.1f composited samples per ray" % (
                        "/C4" if world > 1 else "", args.rays, "f16" if args.half else args.table,
                        ("analytic-scene targets, model conditioned for %d steps, marching its own occupancy grid" % args.condition)
                        if scene else ("random target colours, fixed occupancy=%s (diagnostic state)" % args.regime),
                        rm / total_rays * world, vr / total_rays * world))
        comm_name = args.comm if args.comm != "auto" else ("bf16" if (use_trainer and trainer.grad_comm_dtype == torch.bfloat16) else "f32")
        if world == 1:
            parallelism = "single GPU"
        elif use_trainer and args.shard:
            parallelism = ("ray-sharded dp%d, RCCL reduce-scatter of the %s table gradient -> Adam on the own 1/%d of the table -> all-gather of "
                           "the updated %s, + one 37.6 KB all-reduce [MLP gradient | inf flag]" % (
                               world, "f16" if args.half else comm_name, world,
                               "f16 table copy" if args.half else ("bf16 table copy" if args.table == "bf16" else "f32 table")))
        else:
            parallelism = "ray-sharded dp%d, RCCL all-reduce of one flat %s gradient bucket per step" % (world, comm_name)
        out = {
            "metric": "training rays/sec", "value": total_rays / elapsed, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16-table+f16-mlp" if args.half else ("bf16-table(f32 master)+f16-mlp" if args.table == "bf16" else "f32-table+f16-mlp"), "data": "synthetic",
            "config": {"workload": text, "workload_state": workload,
                       "rays_per_gpu": args.rays, "global_batch": args.rays * world,
                       "parallelism": parallelism,
                       # one or two ~1.1 ms occupancy-update steps in a 20-step window move the average by +-5 %: compare runs by this
                       "grid_updates_in_timed_region": grid_updates,
                       "path": ("FusedTrainer" + ("+hipGraph" if args.graph else "")) if use_trainer else "modules + %s + torch GradScaler / CosineAnnealingLR" % opt_name,
                       "kernel_events_in_timed_region": ((("every step" if ev_every == 1 else "every %d-th step" % ev_every)
                                                          + (", around %s only (the kernel that dominated the warm-up steps, where every "
                                                             "kernel is bracketed)" % sorted(_Probe.only) if _Probe.only else ""))
                                                         if (bool(event_pool) or not use_trainer) else False)},
            "samples_per_sec": rm * world / elapsed, "rm_samples_per_ray": rm / total_rays * world, "vr_samples_per_ray": vr / total_rays * world,
            "live_samples_per_step": live_avg,
            "shaded_samples_last_step": (trainer.shaded_samples() if use_trainer else None),
            "ns_per_live_sample": (elapsed / args.steps * 1e9 / live_avg) if (live_avg and use_trainer) else None,
            "ms_per_step_no_prefetch": None if elapsed_np is None else elapsed_np / args.steps * 1e3,
            "kernels": ks, "critical_path_gaps": gaps, "roofline": roof, "rooflines": rooflines,
        }
        if world > 1:
            out.update(comm_fields(trainer if use_trainer else None, dict(ctx, ms_per_step=out.get("ms_per_step")), args))
        if not args.no_cpu_baseline and world == 1 and not garden:     # the CPU leg restates the C2 workload only
            if scene:
                out["cpu_baseline"] = cpu_baseline(model.density_bitfield.cpu().numpy(), args.cpu_seconds,
                                                   table=model.pos_encoder.hash_table.detach().float().view(-1).cpu().numpy(),
                                                   weights=[w.detach().cpu() for w in model._mlp_weights()], targets="scene")
            else:
                out["cpu_baseline"] = cpu_baseline(bits_np if args.regime == "lego" else np.load(golden)["density_bitfield"],
                                                   args.cpu_seconds)
    return out


if __name__ == "__main__":
    main()
