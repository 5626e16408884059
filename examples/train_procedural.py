"""End-to-end training on a procedural "Lego-shape" scene (there is no dataset and no network in this environment).

An analytic density/colour field (a base plate, a tower and studs inside [-0.35, 0.35]^3, white background) is rendered
to training and test views by dense ray integration in torch; an Instant-NGP model is then trained on it with exactly
the reference's schedule (train.py:54-58,137-201: batch 8192 random pixels over all images, Adam 1e-2 eps 1e-15, cosine
decay to lr/30, GradScaler 2^19, occupancy-grid update every 16 steps with a 256-step warm-up, mark_invisible_cells) and
evaluated with the reference's progressive test-time renderer (rendering.py:62-158).  This is NOT Synthetic-NeRF Lego:
PSNR numbers from it say "the whole path trains and renders", not "matches the paper's scene".

    python examples/train_procedural.py --steps 3000 --path trainer|modules
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ngp_hip.synthetic import procedural_render_gt as render_gt  # noqa: E402  (the analytic scene lives in ngp_hip/synthetic.py)


def cameras(n, radius, seed, device):
    g = torch.Generator().manual_seed(seed)
    z = 0.1 + 0.8 * torch.rand(n, generator=g)
    phi = 2 * math.pi * torch.rand(n, generator=g)
    r = (1 - z * z).sqrt()
    pos = radius * torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
    fwd = F.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = F.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    down = torch.cross(fwd, right, dim=-1)
    return torch.cat([torch.stack([right, down, fwd], -1), pos[..., None]], -1).to(device)     # [n,3,4] c2w (right, down, front)


def pixel_dirs(wh, focal, device):
    ys, xs = torch.meshgrid(torch.arange(wh, device=device), torch.arange(wh, device=device), indexing="ij")
    return torch.stack([(xs - wh / 2 + 0.5) / focal, (ys - wh / 2 + 0.5) / focal, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(-1, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--wh", type=int, default=200)
    ap.add_argument("--n_train", type=int, default=100)
    ap.add_argument("--n_test", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--path", default="trainer", choices=["trainer", "modules"])
    ap.add_argument("--encoder", default="f32", choices=["f32", "bf16", "half"],
                    help="hash table: fp32 (reference default), bf16 storage copy, or the half2 encoder (hash_encoder_half)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(23)
    from modules.networks import NGP
    from modules.rendering import MAX_SAMPLES, render
    from ngp_hip.trainer import FusedTrainer

    focal = 1111.1 * args.wh / 800
    dirs = pixel_dirs(args.wh, focal, dev)
    poses = cameras(args.n_train + args.n_test, 1.39, 23, dev)
    t0 = time.time()
    imgs = torch.stack([render_gt(p[:, 3].expand_as(dirs), dirs @ p[:, :3].T) for p in poses])   # [n, wh*wh, 3]
    torch.cuda.synchronize()
    t_data = time.time() - t0
    train_poses, test_poses = poses[:args.n_train], poses[args.n_train:]
    train_imgs, test_imgs = imgs[:args.n_train], imgs[args.n_train:]

    model = NGP(scale=0.5, max_res=1024, half_opt=args.encoder == "half",
                table_dtype=torch.bfloat16 if args.encoder == "bf16" else None).to(dev)
    K = torch.tensor([[focal, 0, args.wh / 2], [0, focal, args.wh / 2], [0, 0, 1]], device=dev)
    model.mark_invisible_cells(K, train_poses, (args.wh, args.wh))
    if args.path == "trainer":
        trainer = FusedTrainer(model, lr=1e-2, max_steps=args.steps)
    else:
        opt = torch.optim.Adam(model.parameters(), 1e-2, eps=1e-15)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, args.steps, 1e-2 / 30)
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0**19)

    # device-resident training split: batch sampling + get_rays in one kernel (ngp_hip/rays.py)
    from ngp_hip.rays import RayBatcher, get_rays
    batcher = RayBatcher(train_imgs, train_poses, dirs, batch_size=args.batch)
    thr = 0.01 * MAX_SAMPLES / 3**0.5
    torch.cuda.synchronize()
    t0 = time.time()
    log = []
    nxt = batcher.sample()
    for step in range(args.steps):
        cur, nxt = nxt, batcher.sample()                       # one batch of lookahead: its march runs under this step
        rays_o, rays_d, target = cur["rays_o"], cur["rays_d"], cur["rgb"]
        if args.path == "trainer":
            if step % 16 == 0:
                trainer.update_density_grid(thr, warmup=step < 256)
            pre = (nxt["rays_o"], nxt["rays_d"]) if (step + 1) % 16 != 0 else None     # never across a grid update
            st = trainer.step(rays_o, rays_d, target, prefetch=pre)
            if step % 500 == 0:
                log.append((step, trainer.last_loss(), int(st["rm_samples"][0]) / args.batch))
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                if step % 16 == 0:
                    model.update_density_grid(thr, warmup=step < 256)
                res = render(model, rays_o, rays_d, exp_step_factor=0.0)
                loss = F.mse_loss(res["rgb"], target)
            opt.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            sched.step()
            if step % 500 == 0:
                log.append((step, loss.item(), int(res["rm_samples"]) / args.batch))
    torch.cuda.synchronize()
    t_train = time.time() - t0

    psnrs = []
    t0 = time.time()
    with torch.no_grad():
        model.eval()
        for p, gt in zip(test_poses, test_imgs):
            rays_o, rays_d = get_rays(dirs, p)                                                      # fp32, like ray_utils.get_rays
            with torch.autocast("cuda", dtype=torch.float16):
                res = render(model, rays_o, rays_d, test_time=True, exp_step_factor=0.0)
            psnrs.append(-10.0 * math.log10(F.mse_loss(res["rgb"], gt).item()))
    torch.cuda.synchronize()
    out = {"scene": "procedural Lego-shape (NOT Synthetic-NeRF Lego)", "path": args.path, "encoder": args.encoder, "steps": args.steps, "batch": args.batch,
           "train_views": args.n_train, "test_views": args.n_test, "image_wh": args.wh, "test_psnr_mean": sum(psnrs) / len(psnrs),
           "test_psnr_min": min(psnrs), "train_seconds": t_train, "train_rays_per_sec": args.steps * args.batch / t_train,
           "eval_seconds": time.time() - t0, "gt_render_seconds": t_data, "log(step,loss,rm_samples_per_ray)": log,
           "occupied_fraction": float((model.density_bitfield.int().bitwise_and(1) > 0).float().mean())}
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
