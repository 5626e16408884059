"""Write the procedural "Lego-shape" scene (ngp_hip/synthetic.py) to disk in the NSVF Synthetic layout the reference's
loader reads (datasets/nsvf.py:14-110), so that the reference's UNCHANGED train.py can consume it:

    <out>/intrinsics.txt   first token of line 1 = focal length in pixels of the FULL 800x800 image (nsvf.py:37-40)
    <out>/bbox.txt         xmin ymin zmin xmax ymax zmax voxel_size                              (nsvf.py:21-24)
    <out>/pose/{0,1,2}_*.txt   4x4 camera-to-world, [right down front]                            (nsvf.py:78-101)
    <out>/rgb/{0,1,2}_*.png    RGBA, 0_ = train, 1_ = val, 2_ = test                              (nsvf.py:78-90)

Put "Synthetic" and "Lego" in <out> (e.g. .../Synthetic_NSVF_procedural/Lego): the loader then takes the fixed-800-pixel
intrinsics branch and the Lego bound fix-up (scale *= 1.1), exactly as for Synthetic-NeRF Lego.  World coordinates are the
analytic scene's normalised coordinates times 2 * 1.155 * B (B = the bbox half extent written here), which the loader divides
out again (nsvf.py:96-99), so the model sees the scene inside [-0.35, 0.35]^3 of its [-0.5, 0.5]^3 box.

This is NOT Synthetic-NeRF Lego (no dataset, no network in this environment): PSNR numbers on it say "the unchanged driver
trains and evaluates on this package", not "matches the paper's scene".

    python scripts/make_nsvf_scene.py --out /tmp/nsvf/Synthetic_NSVF_procedural/Lego --wh 800 --n_train 100 --n_test 10
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

BBOX_HALF = 1.0                              # bbox.txt = [-B, B]^3
LOADER_SCALE = BBOX_HALF * 1.05 * 1.1        # nsvf.py:25-32 ('Lego' in the path)
WORLD_PER_UNIT = 2.0 * LOADER_SCALE          # world = normalised * this (shift = 0)
SCENE_HALF = 0.36                            # the analytic boxes sit inside [-0.35, 0.35]^3


def cameras(n, radius, seed):
    """n camera-to-world matrices [n,4,4] in NORMALISED coordinates: on the upper hemisphere, looking at the origin."""
    g = torch.Generator().manual_seed(seed)
    z = 0.1 + 0.8 * torch.rand(n, generator=g)
    phi = 2 * math.pi * torch.rand(n, generator=g)
    r = (1 - z * z).sqrt()
    pos = radius * torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
    fwd = F.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = F.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    down = torch.cross(fwd, right, dim=-1)
    c2w = torch.eye(4).repeat(n, 1, 1)
    c2w[:, :3, :3] = torch.stack([right, down, fwd], -1)
    c2w[:, :3, 3] = pos
    return c2w


def garden_cameras(n, seed):
    """n camera-to-world matrices in NORMALISED (= model) coordinates for the Garden-shape scene: a ring of radius 1.0-1.3 at heights
    0.15-0.6 around the object on the table, looking at it (the 360_v2 capture pattern)."""
    g = torch.Generator().manual_seed(seed)
    phi = 2 * math.pi * (torch.arange(n) + torch.rand(n, generator=g)) / n
    rad = 1.0 + 0.3 * torch.rand(n, generator=g)
    pos = torch.stack([rad * torch.cos(phi), rad * torch.sin(phi), 0.15 + 0.45 * torch.rand(n, generator=g)], -1)
    fwd = F.normalize(torch.tensor([0.0, 0.0, -0.1]) + 0.05 * torch.randn(n, 3, generator=g) - pos, dim=-1)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = F.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    down = torch.cross(fwd, right, dim=-1)
    c2w = torch.eye(4).repeat(n, 1, 1)
    c2w[:, :3, :3] = torch.stack([right, down, fwd], -1)
    c2w[:, :3, 3] = pos
    return c2w[torch.randperm(n, generator=g)]                   # (train / test views interleaved around the ring)


def render_views_garden(c2w, wh, focal, device, model_scale, n_samples=768):
    """[n, wh*wh, 3] radiance of the analytic Garden-shape scene (synthetic.garden_field) inside [-model_scale, model_scale]^3,
    black background (what rendering.py composites onto for exp_step_factor > 0)."""
    from ngp_hip.synthetic import garden_render_gt
    ys, xs = torch.meshgrid(torch.arange(wh, device=device), torch.arange(wh, device=device), indexing="ij")
    dirs = torch.stack([(xs - wh / 2 + 0.5) / focal, (ys - wh / 2 + 0.5) / focal, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(-1, 3)
    out = []
    for p in c2w.to(device):
        d = dirs @ p[:3, :3].T
        out.append(garden_render_gt(p[:3, 3].expand_as(d).contiguous(), d.contiguous(), scale=model_scale, n_samples=n_samples).cpu())
    return torch.stack(out)


def render_views(c2w, wh, focal, device, n_samples=512):
    """[n, wh*wh, 3] float32 radiance of the analytic scene, white background.  Only the rays that meet the scene's bounding
    box are integrated (inside that box, n_samples midpoint samples); everything else is background."""
    from ngp_hip.synthetic import procedural_render_gt
    ys, xs = torch.meshgrid(torch.arange(wh, device=device), torch.arange(wh, device=device), indexing="ij")
    dirs = torch.stack([(xs - wh / 2 + 0.5) / focal, (ys - wh / 2 + 0.5) / focal, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(-1, 3)
    out = []
    for p in c2w.to(device):
        d = dirs @ p[:3, :3].T
        o = p[:3, 3].expand_as(d)
        inv = 1.0 / d
        t0, t1 = (-SCENE_HALF - o) * inv, (SCENE_HALF - o) * inv
        hit = torch.maximum(t0, t1).amin(-1) > torch.minimum(t0, t1).amax(-1).clamp_min(0.0)
        img = torch.ones(wh * wh, 3, device=device)
        if hit.any():
            img[hit] = procedural_render_gt(o[hit].contiguous(), d[hit].contiguous(), n_samples=n_samples, scale=SCENE_HALF)
        out.append(img.cpu())
    return torch.stack(out)


def write_scene(out, wh=800, n_train=100, n_val=0, n_test=10, radius=1.39, seed=23, device=None, n_samples=512, scene="lego",
                model_scale=8.0):
    """Returns a small dict describing what was written.  scene="garden": the analytic unbounded scene (object in the unit box the
    bbox file names, ground and far boxes out to +-model_scale: train it with `--scale <model_scale>`, like 360_v2 Garden)."""
    from PIL import Image
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    os.makedirs(os.path.join(out, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(out, "pose"), exist_ok=True)
    focal_full = 1111.111                                        # Synthetic-NeRF intrinsics at 800 px
    focal = focal_full * wh / 800.0
    with open(os.path.join(out, "intrinsics.txt"), "w") as f:
        f.write("%.3f 400.0 400.0 0.\n0. 0. 0.\n1.\n800 800\n" % focal_full)
    with open(os.path.join(out, "bbox.txt"), "w") as f:
        f.write("%.6f %.6f %.6f %.6f %.6f %.6f 0.4\n" % ((-BBOX_HALF,) * 3 + (BBOX_HALF,) * 3))
    n = n_train + n_val + n_test
    c2w = garden_cameras(n, seed) if scene == "garden" else cameras(n, radius, seed)
    t0 = time.time()
    imgs = render_views_garden(c2w, wh, focal, device, model_scale) if scene == "garden" else render_views(c2w, wh, focal, device, n_samples)
    t_render = time.time() - t0
    t0 = time.time()
    for k in range(n):
        split = 0 if k < n_train else (1 if k < n_train + n_val else 2)
        name = "%d_%s_%04d" % (split, ("train", "val", "test")[split], k)
        world = c2w[k].clone().double()
        world[:3, 3] *= WORLD_PER_UNIT
        np.savetxt(os.path.join(out, "pose", name + ".txt"), world.numpy(), fmt="%.10f")
        rgb = (imgs[k].reshape(wh, wh, 3).clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).numpy()
        rgba = np.concatenate([rgb, np.full((wh, wh, 1), 255, np.uint8)], -1)
        Image.fromarray(rgba, "RGBA").save(os.path.join(out, "rgb", name + ".png"), compress_level=1)
    return {"out": out, "image_wh": wh, "downsample_to_pass": wh / 800.0, "n_train": n_train, "n_val": n_val, "n_test": n_test,
            "gt_render_seconds": t_render, "png_write_seconds": time.time() - t0, "device": str(device),
            "scene": ("procedural Garden-shape unbounded scene, model scale %g (ngp_hip/synthetic.py) -- NOT 360_v2 Garden" % model_scale)
                     if scene == "garden" else "procedural Lego-shape (ngp_hip/synthetic.py) -- NOT Synthetic-NeRF Lego"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--wh", type=int, default=800, help="image side; the loader assumes 800 * --downsample, so pass --downsample wh/800 to train.py")
    ap.add_argument("--n_train", type=int, default=100)
    ap.add_argument("--n_val", type=int, default=0)
    ap.add_argument("--n_test", type=int, default=10)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--scene", default="lego", choices=["lego", "garden"])
    ap.add_argument("--model_scale", type=float, default=8.0, help="--scene garden: the --scale train.py will be given")
    args = ap.parse_args()
    import json
    print(json.dumps(write_scene(args.out, args.wh, args.n_train, args.n_val, args.n_test, seed=args.seed, scene=args.scene,
                                 model_scale=args.model_scale)))


if __name__ == "__main__":
    main()
