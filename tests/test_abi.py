"""The C-ABI library builds for gfx950, loads, and exports exactly what include/ngp_hip.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared(name="ngp_hip.h"):
    hdr = open(os.path.join(ROOT, "include", name)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(ngp_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(hip_lib):
    """The boundary header and the experimental header are disjoint, together they are exactly what the ctypes table binds, and the
    library exports every one of them."""
    from ngp_hip import lib
    names, extra = _declared(), _declared("ngp_hip_experimental.h")
    assert len(names) >= 19 and not set(names) & set(extra)
    for n in names + extra:
        assert hasattr(hip_lib, n), "libngp_hip.so does not export %s" % n
    assert sorted(lib.SIGNATURES) == sorted(names + extra), "ngp_hip/lib.py binds a different symbol set than the headers declare"
    assert sorted(lib.EXPERIMENTAL) == extra
    assert hip_lib.ngp_abi_version() == 3


def test_no_default_path_calls_an_experimental_entry_point():
    """Round 6: what include/ngp_hip_experimental.h declares is reached only through a non-default switch, a test or a
    diagnostic.  The drop-in surface (modules/, compat/), the operator layer and the fused render never name one; FusedTrainer names exactly
    the two that sit behind switches that are off by default (NGP_EXPERIMENT comm_overlap=1, `_fold_prologue = False`);
    ngp_hip/p2p.py (reached only through FusedTrainer(exchange="p2p")) names the three of the direct exchange."""
    from ngp_hip import lib
    pkg = os.path.join(ROOT, "taichi-nerfs_amd")
    hits = {}
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "lib.py" or f == "render.hip":
                src = open(os.path.join(dp, f)).read()
                for n in lib.EXPERIMENTAL:
                    if re.search(r"\b%s\b" % n, src):
                        hits.setdefault(os.path.relpath(os.path.join(dp, f), pkg), set()).add(n)
    assert hits == {os.path.join("ngp_hip", "trainer.py"): {"ngp_hash_bwd_sliced_main_adam", "ngp_hash_bwd_sliced_main_levels"},
                    # the direct peer-memory exchange (FusedTrainer(exchange="p2p"), a prototype: default "rccl")
                    os.path.join("ngp_hip", "p2p.py"): {"ngp_p2p_max_peers", "ngp_p2p_push", "ngp_p2p_wait"}}, hits


def test_struct_layout_matches_header(tmp_path):
    """ngp_hash_levels: the ctypes mirror has gcc's layout of include/ngp_hip.h, field by field (round 6: + bwd_plan, the scatter-add's
    task-plan bits that used to be per-thread state of the library)."""
    import subprocess
    from ngp_hip.lib import HashLevels
    names = [f[0] for f in HashLevels._fields_]
    assert names[-1] == "bwd_plan"
    src = tmp_path / "lv.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n  printf("%%zu\\n", sizeof(ngp_hash_levels));\n%s'
                   '  printf("%%u %%u\\n", NGP_BWD_PLAN_DETERMINISTIC, NGP_BWD_PLAN_CONCENTRATED);\n  return 0;\n}\n'
                   % (os.path.join(ROOT, "include", "ngp_hip.h"),
                      "".join('  printf("%%zu\\n", offsetof(ngp_hash_levels, %s));\n' % n for n in names)))
    exe = tmp_path / "lv"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(HashLevels) == 16 + 4 * 16 * 4 + 4
    assert out[1:-2] == [getattr(HashLevels, n).offset for n in names]
    from ngp_hip import lib
    assert out[-2:] == [lib.BWD_PLAN_DETERMINISTIC, lib.BWD_PLAN_CONCENTRATED]


def test_render_args_layout_matches_header(tmp_path):
    """ngp_render_args (the argument block of ngp_render_train_fwd / _bwd): the ctypes mirror has the C compiler's layout -- every field
    at the offset gcc gives it in include/ngp_hip.h."""
    import subprocess
    from ngp_hip.lib import RenderArgs
    names = [f[0] for f in RenderArgs._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n  printf("%%zu\\n", sizeof(ngp_render_args));\n%s  return 0;\n}\n'
                   % (os.path.join(ROOT, "include", "ngp_hip.h"),
                      "".join('  printf("%%zu\\n", offsetof(ngp_render_args, %s));\n' % n for n in names)))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(RenderArgs)
    assert out[1:] == [getattr(RenderArgs, n).offset for n in names]


def test_ops_refuse_cpu_tensors(hip_lib):
    """The product path has no CPU fallback: host tensors are rejected loudly."""
    import torch
    from ngp_hip import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.ray_aabb(torch.zeros(4, 3), torch.ones(4, 3), 0.5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.sh16_fwd(torch.zeros(4, 3))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ngp_hip import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "taichi-nerfs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", "").replace("against the oracle", "") \
                    .replace("the oracle", ""), "%s mentions the oracle" % f
