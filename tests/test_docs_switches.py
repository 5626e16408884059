"""INTEGRATION.md section 5 lists every NGP_* environment switch the host code and the C entry points read (a switch nobody
documented is a behaviour nobody can reproduce)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_every_environment_switch_is_documented():
    pat = re.compile(r"(?:environ\.get\(|environ\[|getenv\()\s*\"(NGP_[A-Z0-9_]+)\"")
    used = set()
    for path in _sources():
        used |= set(pat.findall(open(path, errors="replace").read()))
    assert len(used) > 20, used                                   # the scan itself works
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc[doc.index("## 5. Environment switches"):]
    missing = sorted(v for v in used if v not in section and not any(v.startswith(p[:-1]) for p in re.findall(r"`(NGP_[A-Z_]+\*)`", section)))
    assert not missing, "undocumented switches: %s" % missing
