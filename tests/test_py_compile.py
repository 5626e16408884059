"""Every Python source of the repo byte-compiles (the CPU tier imports only part of the package: a syntax error in a GPU-only module
-- ngp_hip/trainer.py, bench.py, the profiling scripts -- would otherwise first show up on the GPU box)."""
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources():
    out = []
    for top in ("taichi-nerfs_amd", "profiles", "scripts", "tests", "oracle", "examples"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            if "_ref" in d or "__pycache__" in d:
                continue
            out += [os.path.join(d, f) for f in files if f.endswith(".py")]
    return out + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]


@pytest.mark.parametrize("path", _sources(), ids=lambda p: os.path.relpath(p, ROOT))
def test_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)
