"""INTEGRATION.md section 5 lists every NGP_* environment switch the host code and the C entry points read (a switch nobody
documented is a behaviour nobody can reproduce) -- and, since round 6, the handful is SHORT: everything that is an A/B knob lives
behind the one NGP_EXPERIMENT string, whose keys are declared in ngp_hip/experiment.py, documented, and enumerated here."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-nerfs_amd"))

# what a user may want to touch (the rest: NGP_EXPERIMENT)
NAMED = {"NGP_FUSED_RENDER", "NGP_FUSED_MLP", "NGP_FUSED_EVAL", "NGP_FUSED_OCCUPANCY", "NGP_HASH_BWD", "NGP_DETERMINISTIC", "NGP_NO_APEX",
         "NGP_HIPCC_EXTRA", "NGP_EXPERIMENT"}


def _sources():
    for base, exts in (("taichi-nerfs_amd", (".py", ".hip", ".h")), (".", (".py",))):
        top = os.path.join(ROOT, base)
        for d, dirs, files in os.walk(top):
            dirs[:] = [x for x in dirs if x not in ("__pycache__", ".git", "gpurun_out", "profiles", "tests", "oracle", "scripts")]
            if base == "." and d != top:
                continue
            for f in files:
                if f.endswith(exts):
                    yield os.path.join(d, f)


def _section():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return doc[doc.index("## 5. Environment switches"):]


def test_every_environment_switch_is_documented_and_the_list_is_short():
    pat = re.compile(r"(?:environ\.get\(|environ\[|getenv\()\s*\"(NGP_[A-Z0-9_]+)\"")
    used = set()
    for path in _sources():
        used |= set(pat.findall(open(path, errors="replace").read()))
    package = {v for v in used if not v.startswith("NGP_BENCH_")}          # (bench.py's own harness switches are documented as a group)
    assert package <= NAMED, "a new named switch (put it behind NGP_EXPERIMENT): %s" % sorted(package - NAMED)
    section = _section()
    missing = sorted(v for v in used if v not in section and not any(v.startswith(p[:-1]) for p in re.findall(r"`(NGP_[A-Z_]+\*)`", section)))
    assert not missing, "undocumented switches: %s" % missing


def test_every_experiment_key_is_declared_documented_and_enumerable(monkeypatch):
    from ngp_hip import experiment
    read = set()
    for path in _sources():
        src = open(path, errors="replace").read()
        read |= set(re.findall(r"_exp\.(?:get|has)\(\s*\"([a-z0-9_]+)\"", src))
        read |= set(re.findall(r"ngp_experiment\(\s*\"([a-z0-9_]+)\"", src))
        read |= set(re.findall(r"_exp\.has\(k\) for k in \(([^)]*)\)", src) and re.findall(r"\"([a-z_]+)\"", " ".join(re.findall(r"_exp\.has\(k\) for k in \(([^)]*)\)", src))))
    assert len(read) > 25, read
    assert read == set(experiment.KEYS), (sorted(read - set(experiment.KEYS)), sorted(set(experiment.KEYS) - read))
    section = _section()
    assert all(("`%s`" % k) in section for k in experiment.KEYS), [k for k in experiment.KEYS if ("`%s`" % k) not in section]
    # the parser: items separated by ';', values may hold commas, unknown keys are errors
    monkeypatch.setenv("NGP_EXPERIMENT", "flush_adam=0; march_shape=4,82944 ;comm_groups=12,8,0")
    assert experiment.parse() == {"flush_adam": "0", "march_shape": "4,82944", "comm_groups": "12,8,0"}
    assert experiment.get("flush_adam") == "0" and experiment.get("prefetch_at", "3") == "3" and experiment.has("march_shape")
    monkeypatch.setenv("NGP_EXPERIMENT", "flash_adam=0")
    import pytest
    with pytest.raises(ValueError):
        experiment.get("flush_adam")


def test_c_side_parser_agrees(tmp_path, monkeypatch):
    """csrc/ngp_device.h: ngp_experiment() -- compiled for the host with gcc and asked for the same keys."""
    import subprocess
    src = tmp_path / "e.cpp"
    hdr = open(os.path.join(ROOT, "taichi-nerfs_amd", "csrc", "ngp_device.h")).read()
    body = hdr[hdr.index("static inline const char* ngp_experiment"):]
    body = body[:body.index("\n}\n") + 3]
    src.write_text("#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n" + body +
                   "int main(int argc, char** argv) { for (int i = 1; i < argc; ++i) { const char* v = ngp_experiment(argv[i]); printf(\"%s\\n\", v ? v : \"<null>\"); } return 0; }\n")
    exe = tmp_path / "e"
    subprocess.run(["g++", "-std=c++17", "-o", str(exe), str(src)], check=True)
    env = dict(os.environ, NGP_EXPERIMENT="bwd_rep_target=64; prep_batch=3;march_shape=4,0;mlp_bwd=reg")
    out = subprocess.run([str(exe), "bwd_rep_target", "prep_batch", "march_shape", "mlp_bwd", "bwd_rep", "hash_fwd_v1"], env=env, check=True,
                         capture_output=True, text=True).stdout.split()
    assert out == ["64", "3", "4,0", "reg", "<null>", "<null>"]
