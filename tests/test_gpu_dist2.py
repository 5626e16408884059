"""The code path `bench.py --gpus N` runs, with TWO ranks: two processes share GPU 0 over gloo (RCCL refuses two
ranks on one device), each runs FusedTrainer on its `shard_rays` half of one global batch.  Checked, for the sharded optimizer
(reduce-scatter -> Adam on the own shard -> all-gather; the default) and for round 1's single all-reduce, f32 and half2
encoders:
  * the exchanged gradient bucket == the single-rank full-batch gradient (compute_gradients, rel 1e-5; f16 gradients: 2e-2);
  * an inf gradient on ONE rank makes BOTH ranks skip the step and back the loss scale off;
  * after 5 optimisation steps spanning an occupancy-grid update, table, MLP weights, 16-bit table copy and bitfield are
    bit-identical across the ranks (and they moved)."""
import os
import sys
import traceback

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, kind, shard_opt, bits_np, out_dir, overlap=None):
    try:
        if overlap:                          # round 4: the exchange overlapped with the scatter-add, one launch per level group
            os.environ["NGP_EXPERIMENT"] = "comm_overlap=1;comm_groups=%s" % overlap
        for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from modules.networks import NGP
        from ngp_hip import synthetic
        from ngp_hip.dist import shard_rays
        from ngp_hip.trainer import FusedTrainer
        dev = torch.device("cuda", 0)
        n = 2048 if world == 2 else 2049         # (equal ray shards: the exchange averages)

        def make():
            torch.manual_seed(0)
            m = NGP(scale=0.5, max_res=1024, half_opt=kind == "half", table_dtype=torch.bfloat16 if kind == "bf16" else None).to(dev)
            m.density_bitfield.copy_(torch.from_numpy(bits_np).to(dev))
            with torch.no_grad():
                m.pos_encoder.hash_table.mul_(0.2 if kind != "half" else 1.0)
            return m
        o, d = synthetic.lego_rays(n, seed=9)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        g = torch.Generator().manual_seed(1)
        target = torch.rand(n, 3, generator=g).to(dev)
        noise = torch.rand(n, generator=g).to(dev)
        a, b = shard_rays(n, rank, world)
        scale0 = 2.0**10 if kind == "half" else 2.0**15
        tr = FusedTrainer(make(), world_size=world, init_scale=scale0, shard_optimizer=shard_opt)
        assert tr.shard == bool(shard_opt) and tr.rank == rank
        assert (tr._groups is not None) == bool(overlap)
        if overlap and world == 3:               # 4 * world = 12 does not divide the groups' lengths: padded staging buffers
            assert not all(g.aligned for g in tr._groups)

        # (1) exchanged gradients == single-rank full-batch gradients
        out = tr.compute_gradients(o[a:b], d[a:b], target[a:b], noise=noise[a:b].contiguous())
        ref = FusedTrainer(make(), world_size=1, init_scale=scale0).compute_gradients(o, d, target, noise=noise)
        assert int(ref["rm_samples"][0]) > 10000

        def rel(x, y):
            return float((x - y).norm() / y.norm())
        tol = 2e-2 if kind == "half" else (1e-5 if world == 2 else 2e-4)    # (3 ranks: 1 / (3 n) and the mean of three are not exact in f32)
        assert rel(out["table_grad"], ref["table_grad"]) < tol, rel(out["table_grad"], ref["table_grad"])
        assert rel(out["mlp_grad"], ref["mlp_grad"]) < (1e-3 if kind == "half" else tol), rel(out["mlp_grad"], ref["mlp_grad"])

        # (2) inf on rank 1 only -> both ranks skip
        before = tr.table.clone()
        bad = target[a:b].clone()
        if rank == 1:
            bad[:, 0] = float("inf")                              # (every ray: a single poisoned ray may well march no sample)
        tr.step(o[a:b], d[a:b], bad, noise=noise[a:b].contiguous())
        c = tr.counters()
        assert c["skipped"] == 1 and c["opt_steps"] == 0, c
        assert tr.loss_scale() == scale0 / 2
        assert torch.equal(before, tr.table)
        assert float(tr.table_grad_store.float().abs().max()) == 0.0

        # (3) 5 steps spanning a grid update: replicas stay bit-identical
        start = tr.table.clone()
        ro, rd = o[a:b].contiguous(), d[a:b].contiguous()
        hits0 = tr.prefetch_hits
        for i in range(14, 19):
            if i % 16 == 0:
                tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=False)
            nxt = (ro, rd) if (i + 1) % 16 != 0 else None         # the next step's march prefetched under this step's exchange
            tr.step(ro, rd, target[a:b], prefetch=nxt)            # (position 4; rank-local jitter noise, like bench.py)
        tr.sync_master()
        assert tr.prefetch_hits - hits0 == 3 and tr._prefetch_at == 4      # steps 15, 17, 18 consumed the side-stream march
        assert not torch.equal(start, tr.table) and tr.counters()["opt_steps"] == 5
        items = {"table": tr.table, "mlp": tr.mlp_flat, "bits": tr.model.density_bitfield, "grid": tr.model.density_grid,
                 "wpack": tr.wpack.view(torch.int16).int(), "state_f": tr.state_f, "state_i": tr.state_i}
        if tr.copy16_store is not None:
            items["copy16"] = tr.copy16_store[:tr.nt].view(torch.int16).int()       # (gloo has no 16-bit integer type)
        for name, t in items.items():
            mine = t.detach().cpu().contiguous()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert all(torch.equal(both[0], x) for x in both[1:]), "replicas differ in %s" % name
        if tr.copy16_store is not None:                           # the 16-bit copy is the cast of the (synced) master
            want = tr.table.half() if kind == "half" else tr.table.bfloat16()
            assert torch.equal(want.view(torch.int16), tr.copy16_store[:tr.nt].view(torch.int16))
        # (4) checkpoint / resume: state_dict() gathers the Adam moments of every shard (each rank only updates its own
        # 1/world) and saves them unpadded; a fresh 2-rank trainer resumed from RANK 0's checkpoint continues like the original
        sd = tr.state_dict()
        model_sd = {k: v.clone() for k, v in tr.model.state_dict().items()}
        assert sd["table_m"].numel() == tr.nt and sd["table_v"].numel() == tr.nt
        half_nt = tr.nt // 2
        for k in ("table_m", "table_v"):
            assert float(sd[k][:half_nt].abs().max()) > 0 and float(sd[k][half_nt:].abs().max()) > 0, "moments of one shard missing in " + k
            if tr._groups is not None:                              # every rank's chunk of every level group arrived
                for g_ in tr._groups:
                    for r_ in range(world):
                        lo_, hi_ = g_.a + r_ * g_.c, min(g_.a + (r_ + 1) * g_.c, g_.b, tr.nt)
                        if hi_ - lo_ > 4096 and g_.l0 >= 8:       # (fine levels: every chunk sees gradients in 5 steps)
                            assert float(sd[k][lo_:hi_].abs().max()) > 0, (k, g_.index, r_)
            mine = sd[k].cpu().contiguous()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert all(torch.equal(both[0], x) for x in both[1:]), "rank checkpoints differ in " + k
        objs = [(sd, model_sd) if rank == 0 else None]
        dist.broadcast_object_list(objs, src=0)                    # everybody resumes from rank 0's files
        sd0, model_sd0 = objs[0]
        m2 = make()
        m2.load_state_dict({k: v.to(dev) for k, v in model_sd0.items()})
        tr2 = FusedTrainer(m2, world_size=world, init_scale=scale0, shard_optimizer=shard_opt)
        tr2.load_state_dict({k: v.to(dev) for k, v in sd0.items()})
        assert torch.equal(tr2.table_m[:tr.nt], tr.table_m[:tr.nt]) and torch.equal(tr2.table, tr.table) and tr2.counters() == tr.counters()
        if tr.copy16_store is not None:
            assert torch.equal(tr2.copy16_store[:tr.nt].view(torch.int16), tr.copy16_store[:tr.nt].view(torch.int16))
        nz = noise[a:b].contiguous()
        for t_ in (tr, tr2):
            for _ in range(2):
                t_.step(ro, rd, target[a:b], noise=nz)
            t_.sync_master()
        # same inputs, same state: the continuation matches up to the run-to-run float-atomic order of the MLP weight gradients
        assert rel(tr2.table.float(), tr.table.float()) < (2e-3 if kind == "half" else 1e-5), rel(tr2.table.float(), tr.table.float())
        assert rel(tr2.mlp_flat, tr.mlp_flat) < 1e-4
        # a 1-rank trainer loads the same checkpoint (no world-dependent padding in the file)
        m3 = make()
        m3.load_state_dict({k: v.to(dev) for k, v in model_sd0.items()})
        tr3 = FusedTrainer(m3, world_size=1, init_scale=scale0)
        tr3.load_state_dict({k: v.to(dev) for k, v in sd0.items()})
        assert torch.equal(tr3.table_v[:tr.nt], sd0["table_v"].to(dev))
        dist.barrier()
        dist.destroy_process_group()
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    except Exception:
        open(os.path.join(out_dir, "fail_%d" % rank), "w").write(traceback.format_exc())
        raise


@pytest.mark.parametrize("kind,shard_opt,overlap,world", [("f32", True, None, 2), ("f32", False, None, 2), ("half", True, None, 2),
                                                          ("half", False, None, 2), ("bf16", True, None, 2),
                                                          ("f32", True, "12,8,0", 2), ("bf16", True, "8,0", 2), ("f32", True, "8,0", 3)])
def test_two_ranks_on_one_gpu(hip_lib, lego_bitfield, tmp_path, kind, shard_opt, overlap, world):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() + hash((kind, shard_opt, overlap, world))) % 300
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, shard_opt, lego_bitfield, str(tmp_path), overlap)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    fails = [f for f in os.listdir(tmp_path) if f.startswith("fail_")]
    msg = "\n".join(open(os.path.join(tmp_path, f)).read() for f in fails)
    assert not fails, msg
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(os.listdir(tmp_path)) == ["ok_%d" % r for r in range(world)]


def _comm_worker(rank, world, port, out_dir):
    """Train the analytic scene for 240 steps on two ranks twice -- fp32 and bf16 gradient transport -- from the same initial model."""
    try:
        for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from modules.networks import NGP
        from ngp_hip import synthetic
        from ngp_hip.trainer import FusedTrainer
        dev = torch.device("cuda", 0)
        n = 4096
        batches = []
        for b in range(8):
            o, d = synthetic.lego_rays(n, seed=300 + 17 * b + rank)              # rank-dependent shards, like bench.py
            o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            batches.append((o, d, synthetic.procedural_render_gt(o, d).contiguous()))
        eval_o, eval_d = synthetic.lego_rays(n, seed=999)
        eval_o, eval_d = torch.from_numpy(eval_o).to(dev), torch.from_numpy(eval_d).to(dev)
        eval_t = synthetic.procedural_render_gt(eval_o, eval_d).contiguous()
        losses = {}
        for comm in (torch.float32, torch.bfloat16):
            torch.manual_seed(0)
            m = NGP(scale=0.5, max_res=1024).to(dev)
            tr = FusedTrainer(m, world_size=world, grad_comm_dtype=comm, lr=1e-2, max_steps=2000)
            assert (tr._comm is not None) == (comm == torch.bfloat16)
            for i in range(240):
                if i % 16 == 0:
                    tr.update_density_grid(0.01 * 1024 / 3**0.5, warmup=True)
                torch.manual_seed(1000 + i)                                      # same jitter noise in both runs
                tr.step(*batches[i % 8])
            tr.sync_master()
            noise = torch.zeros(n, device=dev)
            out = tr.compute_gradients(eval_o, eval_d, eval_t, noise=noise)      # forward on a held-out batch (no optimizer step)
            losses[str(comm)] = float(out["sq_err"].sum()) / (3.0 * n)
            assert tr.counters()["skipped"] == 0
        l32, l16 = losses["torch.float32"], losses["torch.bfloat16"]
        assert l32 < 0.03, l32                                                   # it learned the scene (initial MSE ~0.1)
        assert abs(l16 - l32) < 0.25 * l32, (l32, l16)                           # 8 mantissa bits on the wire do not change where it gets to
        both = [None, None]
        dist.all_gather_object(both, (l32, l16))
        assert both[0] == both[1], both                                          # replicas agree bit for bit in both modes
        dist.barrier()
        dist.destroy_process_group()
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("%r" % (losses,))
    except Exception:
        open(os.path.join(out_dir, "fail_%d" % rank), "w").write(traceback.format_exc())
        raise


def test_bf16_gradient_transport_converges_like_f32(hip_lib, tmp_path):
    """`bench.py --comm bf16` (FusedTrainer(grad_comm_dtype=torch.bfloat16)) halves the reduce-scatter bytes; two
    ranks trained on the analytic scene with it reach the held-out loss of the fp32 exchange (within 25 %), with identical replicas."""
    import torch.multiprocessing as mp
    port = 29950 + os.getpid() % 40
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    fails = [f for f in os.listdir(tmp_path) if f.startswith("fail_")]
    msg = "\n".join(open(os.path.join(tmp_path, f)).read() for f in fails)
    assert not fails, msg
    assert all(p.exitcode == 0 for p in procs) and sorted(os.listdir(tmp_path)) == ["ok_0", "ok_1"]
