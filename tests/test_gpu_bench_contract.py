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
           "--kernel-events-every", "2"]
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
