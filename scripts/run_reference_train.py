"""Run the reference's UNCHANGED driver (train.py + opt.py + gui.py + datasets/) against THIS repo's `modules` package.

The driver files are taken byte for byte from a reference checkout (--ref, default /root/reference) or, where that does not
exist (the GPU box), from `ref_lease.tgz` at the repo root -- a tarball of exactly those files that scripts/make_ref_lease.sh
makes right before a gpurun call and removes afterwards (it is git-ignored: reference sources never enter this repo's history).
They are copied into a scratch "lease" directory WITHOUT the reference's `modules/`; `python <lease>/train.py` then resolves
`modules` (and the import-only stand-ins for taichi / kornia / cv2 / imageio / torchmetrics, which this image does not ship)
from PYTHONPATH = taichi-nerfs_amd : taichi-nerfs_amd/compat.  train.py's sha256 is checked against the value recorded from
/root/reference in this build container, so "unchanged" is verified where the run happens.

The scene is the procedural Lego-shape scene written in the NSVF Synthetic layout (scripts/make_nsvf_scene.py) -- there is no
Synthetic-NeRF Lego in this environment.  Recipe = scripts/train_nsvf_lego.sh:7-11 of the reference.

    python scripts/run_reference_train.py --out profiles/r03_reference_train_py.json            # 20000 steps, 800x800
    python scripts/run_reference_train.py --max_steps 200 --wh 200 --n_train 10 --n_test 2      # the -m gpu smoke test
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tarfile
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_FILES = ["train.py", "opt.py", "gui.py", "datasets"]
# sha256 of /root/reference/{train,opt,gui}.py as this build container holds them (taichi-dev/taichi-nerfs snapshot of SURVEY.md)
EXPECTED_SHA256 = {
    "train.py": "e0e729ce69d6ffaf3adf0556e672c735fede951d497873b8ea7f2ca8df48081b",
    "opt.py": "6bd78e2059e20570b10a203b80e173f26e88a0ce2b3a225307feb3afe9820343",
    "gui.py": "a145ce2374df681f08aa3803124074a7112f6b3fb199967a506715caf0414a06",
}


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def make_lease(ref, lease):
    """Copy the driver files (never modules/) from a checkout or from ref_lease.tgz into `lease`."""
    os.makedirs(lease, exist_ok=True)
    if ref and os.path.exists(os.path.join(ref, "train.py")):
        for name in DRIVER_FILES:
            src = os.path.join(ref, name)
            if os.path.isdir(src):
                shutil.copytree(src, os.path.join(lease, name), ignore=shutil.ignore_patterns("__pycache__"))
            else:
                shutil.copy(src, os.path.join(lease, name))
        source = ref
    else:
        tgz = os.path.join(ROOT, "ref_lease.tgz")
        if not os.path.exists(tgz):
            raise SystemExit("no reference checkout at %r and no %s (make it with scripts/make_ref_lease.sh)" % (ref, tgz))
        with tarfile.open(tgz) as t:
            members = [m for m in t.getmembers() if m.name.split("/")[0] in DRIVER_FILES and "__pycache__" not in m.name]
            t.extractall(lease, members=members)
        source = tgz
    assert not os.path.exists(os.path.join(lease, "modules")), "the lease must not contain the reference's modules/"
    shas = {n: sha256(os.path.join(lease, n)) for n in EXPECTED_SHA256}
    bad = {n: s for n, s in shas.items() if s != EXPECTED_SHA256[n]}
    if bad:
        raise SystemExit("driver files differ from the recorded reference snapshot: %r" % bad)
    return source, shas


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--data", default=None, help="scene directory (made if missing); must contain 'Synthetic' and 'Lego'")
    ap.add_argument("--wh", type=int, default=800)
    ap.add_argument("--n_train", type=int, default=100)
    ap.add_argument("--n_test", type=int, default=10)
    ap.add_argument("--max_steps", type=int, default=20000)
    ap.add_argument("--batch_size", type=int, default=8192)
    ap.add_argument("--scene", default="lego", choices=["lego", "garden"],
                    help="garden: the analytic unbounded scene + the recipe of scripts/train_360_v2_garden.sh:5-11 (--scale 8 --batch_size 4096 "
                         "--downsample 0.25) through the NSVF loader (there is no colmap capture here)")
    ap.add_argument("--scale", type=float, default=None, help="train.py --scale (default: 0.5 for lego, 8 for garden)")
    ap.add_argument("--extra", default="", help="extra train.py arguments, e.g. '--distortion_loss_w 1e-3' or '--half_opt'")
    ap.add_argument("--env", default="", help="extra environment, e.g. 'NGP_FUSED_RENDER=0'")
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None, help="write train.py's own stdout (its step / evaluation lines) to this file, verbatim")
    ap.add_argument("--keep", action="store_true", help="keep the scratch work directory")
    ap.add_argument("--cprofile", default=None, help="run train.py under cProfile and write the top of the table (by own time) here")
    args = ap.parse_args()

    work = tempfile.mkdtemp(prefix="ngp_ref_train_")
    lease = os.path.join(work, "lease")
    source, shas = make_lease(args.ref, lease)
    garden = args.scene == "garden"
    scale = args.scale if args.scale is not None else (8.0 if garden else 0.5)
    data = args.data or os.path.join(tempfile.gettempdir(), "ngp_nsvf_%s_%d_%d_%d" % (args.scene, args.wh, args.n_train, args.n_test),
                                     "Synthetic_NSVF_procedural" + ("_garden" if garden else ""), "Lego")
    scene = None
    if not os.path.exists(os.path.join(data, "bbox.txt")):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import make_nsvf_scene
        scene = make_nsvf_scene.write_scene(data, wh=args.wh, n_train=args.n_train, n_test=args.n_test, scene=args.scene, model_scale=scale)
    prof = os.path.join(work, "train.prof")
    cmd = [sys.executable] + (["-m", "cProfile", "-o", prof] if args.cprofile else []) + [os.path.join(lease, "train.py"), "--root_dir", data, "--exp_name", "Lego", "--batch_size", str(args.batch_size),
           "--lr", "1e-2", "--gpu", "0", "--max_steps", str(args.max_steps)]
    if args.wh != 800:
        cmd += ["--downsample", repr(args.wh / 800.0)]
    if scale != 0.5:
        cmd += ["--scale", repr(scale)]
    cmd += args.extra.split()
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "taichi-nerfs_amd"), os.path.join(ROOT, "taichi-nerfs_amd", "compat")])
    for kv in args.env.split():
        k, v = kv.split("=", 1)
        env[k] = v
    t0 = time.time()
    p = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True)
    wall = time.time() - t0
    log = p.stdout
    if args.log:
        os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
        with open(args.log, "w") as f:
            f.write("$ " + " ".join(cmd).replace(lease, "<lease>") + "\n" + log)
    sys.stderr.write(p.stderr[-4000:])
    if args.cprofile and os.path.exists(prof):
        import io
        import pstats
        buf = io.StringIO()
        st = pstats.Stats(prof, stream=buf)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumulative").print_stats(45)
        with open(args.cprofile, "w") as f:
            f.write(buf.getvalue().replace(lease, "<lease>").replace(ROOT, "<repo>"))
    if p.returncode != 0:
        sys.stdout.write(log[-4000:])
        raise SystemExit("train.py exited with %d" % p.returncode)
    steps = re.findall(r"elapsed_time=([0-9.]+)s \| step=(\d+) \| psnr=([0-9.naninf-]+) \| loss=([0-9.e+-]+) \| rays=(\d+) \| rm_s=([0-9.]+) \| vr_s=([0-9.]+)", log)
    ev = re.search(r"evaluation: psnr_avg=([0-9.naninf-]+) \| ssim_avg=([0-9.naninf-]+)", log)
    loaded = "import modules, ngp_hip.lib as l, sys; print(modules.__file__); print(l.LIB_PATH)"
    where = subprocess.run([sys.executable, "-c", loaded], cwd=work, env=env, capture_output=True, text=True).stdout.split()
    last = steps[-1] if steps else None
    out = {
        "what": "the reference's unchanged train.py (+ opt.py, gui.py, datasets/) run against this repo's modules package",
        "scene": ("procedural Garden-shape unbounded scene in NSVF Synthetic layout (scripts/make_nsvf_scene.py --scene garden) -- NOT 360_v2 Garden"
                  if garden else "procedural Lego-shape scene in NSVF Synthetic layout (scripts/make_nsvf_scene.py) -- NOT Synthetic-NeRF Lego"),
        "command": " ".join(cmd).replace(lease, "<lease>"), "pythonpath": "taichi-nerfs_amd:taichi-nerfs_amd/compat", "extra_env": args.env,
        "driver_source": source, "driver_sha256": shas, "driver_sha256_matches_reference_snapshot": True,
        "train_py_diff_vs_reference": "empty (sha256 equal)",
        "modules_resolved_to": where[0] if where else None, "libngp_hip": where[1] if len(where) > 1 else None,
        "image_wh": args.wh, "train_views": args.n_train, "test_views": args.n_test, "max_steps": args.max_steps, "batch_size": args.batch_size,
        "log_lines(elapsed_s,step,psnr,loss,rays,rm_s,vr_s)": [[float(a), int(b), float(c), float(d), int(e), float(f), float(g)] for a, b, c, d, e, f, g in steps],
        "train_seconds": float(last[0]) if last else None,
        "train_rays_per_sec": (int(last[1]) + 1) * args.batch_size / float(last[0]) if last else None,
        "test_psnr_avg": float(ev.group(1)) if ev else None, "test_ssim_avg": float(ev.group(2)) if ev else None,
        "wall_seconds_total": wall, "scene_generation": scene,
        "results_written": sorted(os.listdir(os.path.join(work, "results"))) if os.path.isdir(os.path.join(work, "results")) else [],
    }
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)
    if ev is None:
        raise SystemExit("train.py finished without an evaluation line")


if __name__ == "__main__":
    main()
