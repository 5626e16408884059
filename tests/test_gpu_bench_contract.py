"""bench.py's output contract (one JSON line on stdout; metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload + the `roofline` and `cpu_baseline` objects) on a short
run of the real thing: a fresh process, a small conditioning, 6 timed steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(hip_lib):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--condition", "48", "--cpu-seconds", "1",
           "--kernel-events-every", "2", "--no-configs"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout, got %d" % len(lines)
    d = json.loads(lines[0])
    assert d["metric"] == "train_rays_per_sec" or "rays" in d["metric"]
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6           # value IS rays per step / step time
    assert isinstance(d["dtype"], str) and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    for name, rr in d["rooflines"].items():
        assert 0 < rr["frac"] <= 1.0, (name, rr["frac"])                                      # no kernel is priced above its roofline
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert d["value"] > 10 * c["value"]


def test_bench_configs_array(hip_lib):
    """the default N = 1 line carries short runs of the other BASELINE configs."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--condition", "32", "--no-cpu-baseline",
           "--configs-steps", "4"]
    env = dict(os.environ, NGP_BENCH_CONFIG_CONDITION="32")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["grid_updates_in_timed_region"] in (0, 1)
    names = [c["name"] for c in d["configs"]]
    assert names == ["C2-modules-path", "C2-bf16-table", "C5-half2", "C3-garden", "C2-65536-rays", "C2-init-random50"], names
    for c in d["configs"]:
        assert "error" not in c, c
        assert c["value"] > 0 and c["ms_per_step"] > 0, c
        assert abs(c["value"] - c["rays_per_gpu"] / (c["ms_per_step"] * 1e-3)) / c["value"] < 1e-6
        if c["name"] == "C2-modules-path":         # the reference-surface loop (modules + torch optimizer) in every line
            # (the optimizer train.py:143-156 would pick: compat/apex FusedAdam when importable, torch.optim.Adam otherwise)
            assert c["path"].startswith("modules + ") and "Adam" in c["path"] and c["rm_samples_per_ray"] > 0
        else:
            assert c["path"].startswith("FusedTrainer") and 0 < c["frac"] <= 1.0 and c["dominant_kernel"], c
    # round 5: the headline carries its per-live-sample cost at the top level (the driver keeps top-level keys)
    assert d["live_samples_per_step"] > 0 and abs(d["ns_per_live_sample"] - d["ms_per_step"] * 1e6 / d["live_samples_per_step"]) < 1e-6
    assert "deterministic" in d["config"]["workload_state"]["conditioning_mode"]


def test_bench_self_launches_two_ranks(hip_lib):
    """`python bench.py --gpus 2` WITHOUT torchrun starts two ranks by itself (here both on GPU 0 over gloo: a
    functional run of the N > 1 path, not a measurement) and the line carries the communication fields."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--condition", "32",
           "--kernel-events-every", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NGP_BENCH_BACKEND="gloo", NGP_BENCH_ONE_DEVICE="1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * 8192 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6          # whole-job rays/s
    assert d["comm_ms"] is not None and d["comm_ms"] > 0 and d["rccl_ranks"] == 2
    # round 4: exposed communication is MEASURED -- the same steps re-run with the collectives replaced by their local part
    assert d["ms_per_step_comm_stubbed"] > 0 and abs(d["exposed_comm_ms"] - (d["ms_per_step"] - d["ms_per_step_comm_stubbed"])) < 1e-9
    assert d["comm_overlap"] == {"enabled": False, "level_groups": None}
    assert set(d["comm_breakdown_ms"]) == {"reduce_scatter_table_grad", "all_reduce_mlp_grad_and_flag", "all_gather_table"}
    assert d["comm_bytes_per_rank_per_step"]["reduce_scatter_table_grad"] >= 4 * 11420064
    assert "cpu_baseline" not in d
    # round 5: ONE invocation measures every exchange variant in the same process group -- the headline (in-line
    # fp32, sharded optimizer) and, behind it, the overlapped exchange, the 16-bit exchange and the unsharded all-reduce -- each with
    # its communication time, the measured exposed part, the bus bandwidth they imply and DESIGN 7's model beside it
    cfgs = d["configs"]
    # (the overlapped exchange runs last of the collective variants: the one a real node could hang in, the watchdog keeps what came
    # before; the direct peer-memory exchange after it, in CHILD processes of the ranks: a device fault there cannot take the line)
    assert [c["name"] for c in cfgs] == ["inline-f32 (headline)", "bf16-comm+bf16-table", "no-shard-all-reduce", "overlap-8,0", "p2p-direct"]
    assert "configs_incomplete" not in d
    for c in cfgs:
        assert "error" not in c, c
        assert c["comm_ms"] > 0 and c["ms_per_step_comm_stubbed"] > 0 and c["exposed_comm_ms"] is not None and c["bus_bandwidth_GBs"] > 0
        assert set(c["design7_model_at_bus_bandwidth"]) == {"150_GBs", "300_GBs", "450_GBs"}
    assert cfgs[0]["ms_per_step"] == d["ms_per_step"]
    assert any(k.startswith("wait_reduce_scatter_group") for k in cfgs[3]["comm_breakdown_ms"])
    # round 6: the direct peer-memory exchange (two processes on this one GPU: real hipIpc mappings, no xGMI)
    assert {"p2p_reduce_scatter_table_grad", "p2p_all_gather_table", "all_reduce_mlp_grad_and_flag"} <= set(cfgs[4]["comm_breakdown_ms"])
    b_f32 = sum(cfgs[0]["comm_bytes_per_rank_per_step"].values()); b_16 = sum(cfgs[1]["comm_bytes_per_rank_per_step"].values())
    assert 0.45 < b_16 / b_f32 < 0.55
    assert set(cfgs[2]["comm_breakdown_ms"]) == {"all_reduce_flat_bucket"} or "all_reduce" in " ".join(cfgs[2]["comm_breakdown_ms"])


def test_bench_two_ranks_watchdog_keeps_the_headline(hip_lib):
    """A variant that does not finish must not take the headline with it: with a watchdog of 1 s (the first variant's conditioning alone
    takes longer) rank 0 still prints ONE line -- the headline leg, the variants completed so far, and `configs_incomplete` -- and every
    rank exits."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--condition", "32",
           "--kernel-events-every", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NGP_BENCH_BACKEND="gloo", NGP_BENCH_ONE_DEVICE="1", NGP_BENCH_VARIANT_TIMEOUT="1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm_ms"] > 0 and "did not finish" in d["configs_incomplete"]
    assert d["configs"][0]["name"] == "inline-f32 (headline)" and len(d["configs"]) < 5


def test_bench_two_ranks_isolated_leg_may_die(hip_lib):
    """The direct peer-memory exchange runs in child processes of the ranks: children that abort (as a device fault would make them) leave
    ONE complete line -- the headline, every collective variant, and an `error` record for the leg that died."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--condition", "32",
           "--kernel-events-every", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NGP_BENCH_BACKEND="gloo", NGP_BENCH_ONE_DEVICE="1", NGP_BENCH_CHILD_CRASH="1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfgs = d["configs"]
    assert [c["name"] for c in cfgs] == ["inline-f32 (headline)", "bf16-comm+bf16-table", "no-shard-all-reduce", "overlap-8,0", "p2p-direct"]
    assert "configs_incomplete" not in d and d["value"] > 0
    assert all("error" not in c for c in cfgs[:4]) and "child exit code" in cfgs[4]["error"]


def test_bench_two_ranks_overlapped_exchange(hip_lib):
    """NGP_EXPERIMENT comm_overlap=1: the scatter-add in one launch per level group, each group's reduce-scatter in flight under the next
    group's launch (functional run over gloo on one GPU; the line names the grouping and what the step waited for)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--condition", "32",
           "--kernel-events-every", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NGP_BENCH_BACKEND="gloo", NGP_BENCH_ONE_DEVICE="1", NGP_EXPERIMENT="comm_overlap=1;comm_groups=12,8,0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["comm_overlap"] == {"enabled": True, "level_groups": "12,8,0"}
    assert {"wait_reduce_scatter_group0", "wait_reduce_scatter_group2", "wait_all_gather_group1",
            "wait_all_reduce_mlp_grad_and_flag"} <= set(d["comm_breakdown_ms"])
    assert d["ms_per_step_comm_stubbed"] > 0 and d["exposed_comm_ms"] is not None


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NGP_BENCH_ONE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2"], cwd=ROOT, capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode != 0 and "exposes" in (out.stderr + out.stdout)
